import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import kernels as K
dev="cuda:0"; dtype=torch.float16; C=192; M=51200
x=torch.randn(M,C,device=dev).to(dtype)
w1,b1=torch.randn(4*C,C)*C**-0.5, torch.randn(4*C)*0.1
w2,b2=torch.randn(C,4*C)*(4*C)**-0.5, torch.randn(C,device=dev)*0.1
lw,lb,g2=torch.rand(C)+0.5, torch.randn(C)*0.1, torch.rand(C,device=dev)*0.3
w1f,b1f,w2c=(t.to(dev) for t in K.pack_mlp_weights(w1,b1,w2,lw,lb,dtype))
op=K.MlpFusedOp(x,w1f,b1f,w2c,b2,g2,M,C,1e-6)
for _ in range(5): op()
torch.cuda.synchronize()
