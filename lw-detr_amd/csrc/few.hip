// Few-row GEMM / 3x3 convolution for the single-image latency path (round 6):  out = epilogue( A_view(M,K) x W(N,K)^T ),  M <= a few thousand rows.
//
// Replaces lwdetr_gemm's 64 x 64 DMA-ring kernel on the launches of one or two images whose contraction is long - first of all the six 3x3
// convolutions of the projector's C2f block (models/backbone/projector.py:101-132; M = 1600 pixels, N = 128, K = 9 x 128 = 1152): on 64 x 64 tiles
// that is 50 workgroups, each walking 36 dependent DMA -> barrier -> fragment -> MFMA steps - 19.3 us per launch where the matrix work is < 1 us
// (profiles/r5e_*: neither ring depth nor stage depth nor split-K moved it).
//
// Here nothing goes through LDS and nothing waits for a barrier: a workgroup owns 16 rows x 128 columns, its 8 waves one 16 x 16 output tile each, and a
// wave loads the MFMA fragments of a WHOLE third of the contraction (NB = 12-18 k-chunks of 32) straight from L2 into registers before it multiplies
// them - the contraction is three L2 round trips deep instead of 36 ring steps. What makes that load path fast is the weight layout (the lesson of
// lwdetr_vit_block_few, profiles/r6b_*): W arrives FRAGMENT-MAJOR - [N / 16][K / 32][16][32], lwdetr_amd.kernels.pack_frag16 - so a wave's weight load is
// one contiguous KB; the activation fragment (16 rows x 64 bytes) is the same for the 8 waves of the workgroup and comes out of L1 for seven of them.
// 100 workgroups at one 640 x 640 image (1600 / 16), 800 waves: every CU has work.
// Arithmetic: 16x16x32 MFMAs in k order, f32 accumulation, the epilogue's operation order is lwdetr_gemm's (bias, activation, scale * gamma, residual).
#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_few_zero[4];      // source of the fragments of out-of-image taps

// KCH = Cin / 32 of a 3x3 convolution (a batch = the 3 KCH chunks of one kernel row); PLAIN: KCH = 4, batches of 12 chunks
template <typename T, int AMODE, int KCH>
__global__ __launch_bounds__(512) void gemm_few_kernel(const lwdetr_gemm_desc d) {
    constexpr int NB = 3 * KCH;
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int n_tile = (int)blockIdx.y * (int)(blockDim.x >> 6) + wave;   // this wave's 16 output columns (blockDim = 64 x waves per workgroup)
    if (n_tile * 16 >= d.N) return;                                 // wave-uniform; no barrier anywhere below
    const long m = (long)blockIdx.x * 16 + l15;
    const bool m_ok = m < d.M;
    const long mc = m_ok ? m : d.M - 1;                             // clamped for addressing
    const T* __restrict__ A = (const T*)d.A;
    const T* __restrict__ Wf = (const T*)d.W;
    const T* zero = (const T*)g_few_zero;
    const int nchunks = d.K / 32;
    // row part of the activation address: PLAIN the row itself; CONV3x3 (raster rows, zero padding) the output pixel's image and coordinates
    int pb = 0, py = 0, px = 0;
    if (AMODE == LWDETR_A_CONV3x3) {
        const int hw = d.conv_hout * d.conv_wout;
        pb = (int)(mc / hw);
        const int r = (int)(mc - (long)pb * hw);
        py = r / d.conv_wout; px = r - py * d.conv_wout;
    }
    const T* arow = A + mc * d.lda + g * 8;                          // PLAIN
    const T* wbase = Wf + ((long)n_tile * nchunks * 16 + l15) * 32 + g * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < nchunks; c0 += NB) {                      // wave-uniform trip count
        V8 xa[NB], wa[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = c0 + i < nchunks ? c0 + i : nchunks - 1;  // past the end: a harmless reload, not multiplied
            const T* src;
            if (AMODE == LWDETR_A_CONV3x3) {
                const int ky = c0 / NB, kx = i / KCH, cc = i % KCH;  // batch = kernel row ky (9 KCH chunks in all: never past the end)
                const int iy = py * d.conv_stride + ky - 1, ix = px * d.conv_stride + kx - 1;
                const bool ok = iy >= 0 && iy < d.a_tok.Hp && ix >= 0 && ix < d.a_tok.Wp;
                src = ok ? A + (((long)pb * d.a_tok.Hp + iy) * d.a_tok.Wp + ix) * d.lda + d.a_col0 + cc * 32 + g * 8 : zero;
            } else src = arow + c * 32;
            xa[i] = *(const V8*)src;
            wa[i] = *(const V8*)(wbase + (long)c * 512);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (c0 + i < nchunks) acc = Mma<T>::k32(wa[i], xa[i], acc);       // D[n][row]: lane (row l15, g) holds columns 4 g .. 4 g + 3
    }
    // ---- epilogue: one LINEAR segment
    const lwdetr_gemm_seg& sg = d.seg[0];
    const int n = n_tile * 16 + g * 4;
    f32x4 x = acc;
    if (sg.bias) x += *(const f32x4*)(sg.bias + n);
    const int act = sg.act;
    if (act != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            x[e] = act == ACT_GELU ? gelu_for<T>(x[e]) : (act == ACT_SILU ? x[e] * __builtin_amdgcn_rcpf(1.f + __expf(-x[e])) : (x[e] > 0.f ? x[e] : 0.f));
    }
    x = x * sg.scale;
    if (sg.gamma) x = x * *(const f32x4*)(sg.gamma + n);
    if (!m_ok) return;
    if (sg.res) x += up4<T>(*(const V4*)((const T*)sg.res + m * sg.ldres + n));
    const V4 o = cvt4<T>(x);
    *(V4*)((T*)sg.out + m * sg.ldo + n) = o;
    if (sg.out2) *(V4*)((T*)sg.out2 + m * sg.ld2 + n) = o;
}

template <typename T>
int few_launch(const lwdetr_gemm_desc& d, hipStream_t st) {
    // waves per workgroup = 16-column tiles that share a row tile's activation fragments through L1. The launch is a stream of weights through each
    // CU's load path (a 16-row tile pulls ALL of its columns' weights: 295 KB for the projector's 3x3 convolutions): fewer waves per workgroup spread
    // the same waves over more CUs (LWDETR_GEMM_FEW_WAVES, tuning; measured in profiles/r6c_*)
    int nwv = (int)lwdetr_knob(KNOB_GEMM_FEW_WAVES, 2);
    if (nwv != 1 && nwv != 2 && nwv != 4 && nwv != 8) nwv = 2;
    const int ntile = d.N / 16;
    const dim3 grid((unsigned)((d.M + 15) / 16), (unsigned)((ntile + nwv - 1) / nwv));
    const dim3 block((unsigned)(64 * nwv));
    const int kid = d.a_mode == LWDETR_A_CONV3x3 ? KID_GEMM_CONV : KID_GEMM;
    ProfScope ps(kid, 2.0 * d.M * d.N * d.K, ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N) * sizeof(T), st);
    if (d.a_mode == LWDETR_A_CONV3x3) {
        // a batch = the three taps of one kernel row (3 x Cin / 32 chunks): Cin = 128 -> 12, Cin = 192 -> 18
        if (d.conv_cin == 192) hipLaunchKernelGGL((gemm_few_kernel<T, LWDETR_A_CONV3x3, 6>), grid, block, 0, st, d);
        else hipLaunchKernelGGL((gemm_few_kernel<T, LWDETR_A_CONV3x3, 4>), grid, block, 0, st, d);
    } else hipLaunchKernelGGL((gemm_few_kernel<T, LWDETR_A_PLAIN, 4>), grid, block, 0, st, d);
    return lwdetr_check_launch();
}

}  // namespace

extern "C" int lwdetr_gemm_few(const lwdetr_gemm_desc* desc, int dtype, void* hip_stream) {
    if (!desc) return LWDETR_ERR_BAD_ARG;
    const lwdetr_gemm_desc& d = *desc;
    if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.W || d.nseg != 1 || !d.seg[0].out || d.seg[0].n_begin != 0) return LWDETR_ERR_BAD_ARG;
    if (d.M == 0) return LWDETR_OK;
    const lwdetr_gemm_seg& g = d.seg[0];
    if ((dtype != DT_F16 && dtype != DT_BF16) || d.A2 || d.M > 8192 || d.K % 32 != 0 || d.N % 16 != 0 || g.mode != LWDETR_OUT_LINEAR || g.rowmask || g.ln_stats ||
        g.res_mod > 0 || g.n_end < d.N || g.ldo % 4 != 0 || (g.res && g.ldres % 4 != 0) || (g.out2 && g.ld2 % 4 != 0) || d.lda % 8 != 0 ||
        ((size_t)d.A & 15) != 0 || ((size_t)d.W & 15) != 0 || ((size_t)g.out & 7) != 0)
        return LWDETR_ERR_UNSUPPORTED;
    if (d.a_mode == LWDETR_A_CONV3x3) {
        if (d.a_tok.winmajor || d.conv_cin % 32 != 0 || d.K != 9 * d.conv_cin || d.a_col0 % 8 != 0 || d.conv_hout <= 0 || d.conv_wout <= 0 ||
            (d.conv_stride != 1 && d.conv_stride != 2) || (d.conv_cin != 128 && d.conv_cin != 192))
            return LWDETR_ERR_UNSUPPORTED;
    } else if (d.a_mode != LWDETR_A_PLAIN) return LWDETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)hip_stream;
    return dtype == DT_F16 ? few_launch<f16>(d, st) : few_launch<bf16>(d, st);
}
