// tamcmc_eval.hip -- the hot kernel: for one (chain, tile of frequency bins) build the model
// spectrum M(x) from the chain's multiplet table, fold it into the chi(2,2p) / chi_square
// log-likelihood partial sums and (GRAD) into the per-multiplet gradient partial sums.
//
// Reference functions fused here:
//   build_l_mode_a1etaa3 & friends   build_lorentzian.cpp:15-205   sum over m of (asymmetric) Lorentzians
//   optimum_lorentzian_calc_*        build_lorentzian.cpp:258-769  add the multiplet inside [imin,imax) only
//   harvey_like / harvey1985         noise_models.cpp:17-69
//   model_Test_Gaussian / _Harvey_Gaussian   models.cpp:1968-2034 (Gaussian term)
//   likelihood_chi22p / _chi_square  likelihoods.cpp:17-39
//
// Mapping to gfx950 (wave64):
//   grid = (tiles, chains); workgroup = 256 threads = 4 waves.  A tile is S sub-blocks of 256*KU
//   consecutive bins; thread t owns bins base + (u*KU + k)*256 + t, so every global load is a
//   coalesced 8-byte-per-lane stream (x, y, log x: 24 B per bin; the 2.4 MB working set of a 1e5-bin
//   star stays in L2).  KU bins are processed together (instruction-level parallelism across the
//   reciprocal's latency); the sub-block loop is NOT unrolled, so the register footprint is set by
//   KU while the work per reduction is set by KU*S.
//   The chain's multiplets whose window meets the tile are compacted (ballot, in table order, so the
//   summation order is fixed) and staged once per tile in LDS; every thread walks the staged list
//   with wave-uniform control flow and broadcast LDS reads.
//   Arithmetic per Lorentzian component: d = 2x - 2nu; E = d*d + Gamma^2 (2 fp64 ops); one
//   reciprocal per MULTIPLET via batch inversion (prefix products of the E's), not per component:
//   sum_m h_m Gamma^2 / E_m.  This is algebraically the reference's H V_m / (1 + 4 (x-nu_m)^2/Gamma^2)
//   and differs from it only in rounding (<~1e-15 relative per bin).
//   Harvey profiles: (1e-3 tau x)^p = exp(p (log(1e-3 tau) + log x)) with a log x table; inside a
//   tile exp(p log x) = t_center * exp(z), |z| <= 0.04, by an 8th-degree Taylor polynomial (error
//   < 1e-18); wider tiles in log x fall back to a full exp per bin.
//   Sum of log M: mantissas are multiplied and exponents added per bin (v_frexp_*), one log per
//   thread and tile instead of one per bin.
//   Reductions: likelihood: wave shuffles -> one LDS slot per wave -> one partial per (chain, tile);
//   gradient: a transposing butterfly (V values per lane cost ~V exchanges, not 6V) -> LDS -> one
//   partial per (chain, tile, multiplet, slot).  The sums over tiles run in a fixed order (last-arriving
//   workgroup of the chain, or tamcmc_backward_kernel on the gradient path); the only atomic is an arrival
//   counter, never a floating-point accumulation: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tamcmc_eval_body.h"

template <int KU, bool GRAD>
__global__ __launch_bounds__(TM_THREADS, (GRAD ? TM_LB_GRAD : TM_LB_FWD)) void tamcmc_eval_kernel(TmEvalArgs a)
{
    TM_STAMP(0);
    // Workgroups go to the 8 XCDs round-robin by linear id (slot + tiles * chain).  Tile costs follow the mode
    // pattern of the spectrum, so every XCD should see every tile equally often.  With tile = slot that holds only
    // when tiles = 1 (mod 8); a tile count that is a multiple of 8 pins each tile to one XCD and the XCD with the
    // densest tiles finishes ~15 % late (profiles/README.md).  Rotating the slot -> tile map by r * chain with
    // r = (tiles - 1) mod 8 makes (XCD - tile) = chain (mod 8) for every tile count.
    const int tid = threadIdx.x;
    int chain, tile;
    if (a.order_mode == 0) {
        chain = blockIdx.y;
        const unsigned rn = (unsigned)((a.tiles - 1) & 7) * (unsigned)(chain & 0xffff);          // < 2^19
        const unsigned rq = (unsigned)(((unsigned long long)rn * a.tile_magic) >> 40);           // rn / tiles (scalar unit)
        tile = (int)blockIdx.x + (int)(rn - rq * (unsigned)a.tiles);
        if (tile >= a.tiles) tile -= a.tiles;
    } else {
        chain = blockIdx.x;
        tile = (a.order_mode == 2) ? a.order[(size_t)chain * a.tiles + blockIdx.y] : (int)blockIdx.y;
    }
    // Issue priority by launch rank: the costliest quarter of a chain's tiles runs at priority 3, the next at 2, 1, 0.
    // Waves are served oldest-first anyway; this keeps a late-placed cheap workgroup from slowing the long ones it
    // joins (measured: the opposite assignment costs 4 %, this one gains ~1 %; profiles/README.md).
    if (a.order_mode != 0) {
        const int r4 = (4 * (int)blockIdx.y) / a.tiles;
        if (r4 == 0) __builtin_amdgcn_s_setprio(3); else if (r4 == 1) __builtin_amdgcn_s_setprio(2); else if (r4 == 2) __builtin_amdgcn_s_setprio(1);
    }
    extern __shared__ double s_dyn[];                                // weights of pass 2: [TM_THREADS * KU * S]   (GRAD only)
    tm_eval_body<KU, GRAD>(a, chain, tile, s_dyn);
}

template <int KU>
static int tm_launch_eval_k(const TmEvalArgs &a, int Nchains, bool grad, hipStream_t stream)
{
    dim3 grid(a.tiles, Nchains), block(TM_THREADS);
    if (a.order_mode != 0) grid = dim3(Nchains, a.tiles);
    const int Smax = a.tile_big > a.tile_small ? a.tile_big : a.tile_small;
    size_t lds = grad ? (size_t)TM_THREADS * KU * Smax * sizeof(double) : 8;
    if (lds > 48 * 1024) {
        const void *fn = grad ? reinterpret_cast<const void *>(tamcmc_eval_kernel<KU, true>)
                              : reinterpret_cast<const void *>(tamcmc_eval_kernel<KU, false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
#ifdef TM_TRACE
    static unsigned long long *d_trace = nullptr;
    static size_t cap = 0;
    const size_t nblk = (size_t)a.tiles * Nchains;
    const char *tf = getenv("TAMCMC_TRACE_FILE");
    const bool trace_grad = getenv("TAMCMC_TRACE_GRAD") != nullptr;
    if (tf && grad == trace_grad) {
        if (nblk > cap) { (void)hipFree(d_trace); (void)hipMalloc(&d_trace, nblk * 4 * sizeof(unsigned long long)); cap = nblk; }
        (void)hipMemsetAsync(d_trace, 0, nblk * 4 * sizeof(unsigned long long), stream);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tm_trace), &d_trace, sizeof(d_trace), 0, hipMemcpyHostToDevice, stream);
    }
#endif
    if (grad) hipLaunchKernelGGL((tamcmc_eval_kernel<KU, true>), grid, block, lds, stream, a);
    else      hipLaunchKernelGGL((tamcmc_eval_kernel<KU, false>), grid, block, lds, stream, a);
#ifdef TM_TRACE
    if (tf && grad == trace_grad) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(nblk * 4);
        (void)hipMemcpy(h.data(), d_trace, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        FILE *f = fopen(tf, "wb");
        if (f) { long long dims[2] = {(long long)grid.y, (long long)grid.x}; /* rows x columns of the launch grid */ fwrite(dims, sizeof(dims), 1, f); fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
#endif
    return (int)hipGetLastError();
}

int tm_launch_eval(const TmEvalArgs &a, int Nchains, int KU, bool grad, void *stream)
{
    if (a.n_mult > TM_MAXMULT) return (int)hipErrorInvalidValue;
    switch (KU) {
    case 1: return tm_launch_eval_k<1>(a, Nchains, grad, (hipStream_t)stream);
    case 2: return tm_launch_eval_k<2>(a, Nchains, grad, (hipStream_t)stream);
    case 4: return tm_launch_eval_k<4>(a, Nchains, grad, (hipStream_t)stream);
    default: return (int)hipErrorInvalidValue;
    }
}
