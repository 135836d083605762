"""GPU: kernels of one launch chain beside the kernels of another (LWDETR._forward_chains runs the halves of a dense batch on two
streams, so workgroups of DIFFERENT kernels share CUs - which never happens on one stream).

Background: the decoder's fused sampling kernel, a pure per-lane function of static inputs, returned
different bits in lanes 48-63 of a few waves whenever its waves shared SIMDs with the MFMA waves of the other chain's GEMM /
attention kernels - until msda.hip was compiled without packed-f32 instructions. This test runs EVERY launch of a plan under
that load and compares all buffers of the plan with the launch's result on an idle GPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _two_plans(size, batch, res, dtype, seed=3):
    import lwdetr_amd
    from lwdetr_amd.models import lwdetr as L
    from lwdetr_amd.synth import synth_images, synth_state_dict
    model, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args(size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to("cuda:0").to(dtype).eval()
    x = synth_images(batch, res, res, seed=seed).to("cuda:0").to(dtype)
    try:
        L.set_streams(2)
        model(x)                                     # builds both plans and leaves every buffer holding real data
        torch.cuda.synchronize()
    finally:
        L.set_streams(0)
    part = batch // 2
    return model, x, model._plans[(part, res, res, 0)], model._plans[(part, res, res, 1)]


# BASELINE batches (configs 2, 3 and the per-GPU shard of 4): the kernel selection depends on the row count (large-tile GEMM from
# 16 384 rows, patch-resident convolution from 100 workgroups, one-wave window attention), so the chains are stressed at the
# benchmarked sizes, not at smaller stand-ins
@pytest.mark.parametrize("size,batch,res,dtype", [("small", 32, 640, torch.float16), ("medium", 64, 640, torch.bfloat16),
                                                  ("large", 32, 640, torch.float16)])
def test_every_launch_repeats_bit_for_bit_beside_the_other_chain(size, batch, res, dtype):
    model, x, p0, p1 = _two_plans(size, batch, res, dtype)
    part = batch // 2
    bufs = p1.buffers
    saved = [b.clone() for b in bufs]
    ref = [torch.empty_like(b) for b in bufs]
    side = torch.cuda.Stream()
    ops = list(p1.ops_backbone) + list(p1.ops_enc) + [p1.op_rowmax, p1.op_topk] + list(p1.ops_sel) + list(p1.ops_dec)
    flags = torch.zeros(len(ops), dtype=torch.int32, device="cuda:0")
    solo = torch.zeros(len(ops), dtype=torch.int32, device="cuda:0")

    def trial(op):
        for b, s in zip(bufs, saved):
            b.copy_(s)
        op(side.cuda_stream)

    def count_diff(acc, j):
        for b, r in zip(bufs, ref):
            acc[j] += (b != r).any().to(torch.int32)

    for j, op in enumerate(ops):
        with torch.cuda.stream(side):
            trial(op)
            for b, r in zip(bufs, ref):
                r.copy_(b)
            trial(op)
            count_diff(solo, j)                     # idle GPU: the launch repeats (this also catches a launch that is not a
        torch.cuda.synchronize()                     # function of the plan's buffers alone)
        for rep in range(4):
            p0.run(x[:part])                         # the load: the other part's whole forward on the current stream
            with torch.cuda.stream(side):
                for _ in range(3):
                    trial(op)
                    count_diff(flags, j)
        torch.cuda.synchronize()
    names = [type(o).__name__ for o in ops]
    bad_solo = [(j, names[j], int(v)) for j, v in enumerate(solo.tolist()) if v]
    bad = [(j, names[j], int(v)) for j, v in enumerate(flags.tolist()) if v]
    assert not bad_solo, f"launches that do not repeat on an idle GPU: {bad_solo}"
    assert not bad, f"launches whose result changed beside the other chain (launch index, op, buffer mismatches in 12 trials): {bad}"
