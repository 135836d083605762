import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import kernels as K
dev="cuda:0"; dtype=torch.float32
def rnd(*s, seed): 
    g=torch.Generator().manual_seed(seed); return torch.randn(*s, generator=g).to(dtype).to(dev)
heads,b,hd=3,2,16; twp,tw=12,9; tp=192
q=rnd(b,heads,tp,hd,seed=1); k=rnd(b,heads,tp,hd,seed=2); v=rnd(b,heads,tp,hd,seed=3)
spike = "--nospike" not in sys.argv
if spike: k[0,0,tp-3]=q[0,0,5]*4
scale=K.attention_scale(hd); qs=(q.float()*scale).to(dtype)
out=torch.zeros(b*tp,heads*hd,dtype=dtype,device=dev)
K.AttnOp(qs,k,v.transpose(2,3).contiguous(),out,B=b,heads=heads,hd=hd,Tp=tp,ldo=heads*hd,seqs_per_img=1,seq_tok_stride=tp,keys_per_seq=tp,sub_stride=twp,sub_len=tw,kind=0)()
o=out.reshape(b,tp,heads,hd).permute(0,2,1,3).float()
valid=(torch.arange(tp,device=dev)%twp)<tw
s=(qs.float()/math.log2(math.e))@k.float().transpose(-2,-1); s=s.masked_fill(~valid[None,None,None,:],float("-inf"))
ref=s.softmax(-1)@v.float()
err=(o-ref).abs().amax(-1)   # (b,heads,tp)
err=err*valid
print("spike",spike,"max err",err.max().item())
bad=(err>1e-3).nonzero()
print("n bad",bad.shape[0], bad[:20].tolist())
print("isnan", torch.isnan(o).sum().item())
s2=(qs.float()/math.log2(math.e))@k.float().transpose(-2,-1)
ref_nomask=s2.softmax(-1)@v.float()
print("vs unmasked ref:", ((o-ref_nomask).abs().amax(-1)*valid).max().item())
for sl in (9,8,10,12):
    vv=(torch.arange(tp,device=dev)%twp)<sl
    r=s2.masked_fill(~vv[None,None,None,:],float("-inf")).softmax(-1)@v.float()
    print("sub_len",sl,((o-r).abs().amax(-1)*valid).max().item())
# first step only mask?
for nst in (1,2,3):
    vv=valid.clone(); vv[32*nst:]=True
    r=s2.masked_fill(~vv[None,None,None,:],float("-inf")).softmax(-1)@v.float()
    print("mask only first",nst,"steps:",((o-r).abs().amax(-1)*valid).max().item())
