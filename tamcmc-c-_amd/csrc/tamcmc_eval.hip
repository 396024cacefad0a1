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
//   grid = (chains, tiles), tile-major and costliest-first; workgroup = 256 threads = 4 waves.  The grid is cut into
//   units of 512 bins; a tile is a run of consecutive units (at most 8 on the gradient path, 16 on the likelihood path)
//   -- tiles of EQUAL LENGTH by default (25 per chain at 1e5 bins); with TAMCMC_EQUAL_COST=1 the setup kernel instead
//   chooses a chain's bounds so that its tiles cost about the same (tamcmc_dev.h, tamcmc_setup_body.h; measured: no
//   faster).  Thread t owns bins
//   unit*512 + k*256 + t, so every global load is a coalesced 8-byte-per-lane stream (x, y, log x: 24 B per bin; the
//   2.4 MB working set of a 1e5-bin star stays in L2).  Four bins per thread are processed together (instruction-level
//   parallelism across the reciprocal's latency); the group loop is NOT unrolled, so the register footprint is set by
//   the group while the work per reduction is set by the tile.
//   Everything that is uniform over the workgroup -- the tile header, the list of multiplets whose window meets the
//   tile (table order, so the summation order is fixed), each multiplet's record, the cells' background polynomials --
//   is read through the constant address space: scalar loads into SGPRs, which the VALU instructions take directly as
//   operands.  No LDS staging, no barrier before the arithmetic.  A group outside a multiplet's window skips it with a
//   scalar branch; a group wholly inside it (the usual case) evaluates it without any per-bin test.
//   Arithmetic per Lorentzian component: d = 2x - 2nu; E = d*d + Gamma^2 (2 fp64 ops); the likelihood path builds a
//   whole multiplet as one rational N/Q (3 more ops per component) and takes ONE reciprocal per multiplet; the gradient
//   path needs every 1/E_m and gets them by batch inversion (prefix products of the E's).  Either is algebraically the
//   reference's H V_m / (1 + 4 (x-nu_m)^2/Gamma^2) and differs from it only in rounding (<~1e-15 relative per bin).
//   Harvey profiles: per 4096-bin cell the whole background is one degree-8 polynomial in (log x - log x_c) built by
//   the setup kernel (truncation error < 1e-16 relative, checked there; wider cells in log x fall back to exp per bin).
//   Sum of log M: mantissas are multiplied and exponents added per bin (v_frexp_*), one log per thread and tile
//   instead of one per bin.
//   Reductions: likelihood: wave exchanges (v_permlane*_swap, DPP) -> one LDS slot per wave -> one partial per (chain,
//   tile); gradient: a transposing butterfly (V values per lane cost ~V exchanges, not 6V) -> LDS -> one partial per
//   (chain, tile, multiplet, slot).  The sums over tiles run in a fixed order (last-arriving workgroup of the chain, or
//   tamcmc_backward_kernel on the gradient path); the only atomic is an arrival counter, never a floating-point
//   accumulation: results are bitwise reproducible.
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tamcmc_eval_body.h"

template <bool GRAD, bool GEN>
__global__ __launch_bounds__(TM_THREADS, (GRAD ? TM_LB_GRAD : TM_LB_FWD)) void tamcmc_eval_kernel(TmEvalArgs a)
{
    TM_STAMP(0);
    // Workgroups go to the 8 XCDs round-robin by linear id.  Default (order_mode 2): x = chain, y = launch rank, so
    // consecutive workgroups run the same region of the spectrum for different chains (shared x / y / log x lines stay
    // hot in the XCD's L2) and every XCD gets whole chains.  order_mode 0 (chain-major: x = slot, y = chain) rotates the
    // slot -> tile map by ((tiles - 1) mod 8) * chain so that no tile is pinned to one XCD (profiles/README.md).
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
    if (a.order_mode != 0 && a.prio == 1) {
        const int r4 = (4 * (int)blockIdx.y) / a.tiles;
        if (r4 == 0) __builtin_amdgcn_s_setprio(3); else if (r4 == 1) __builtin_amdgcn_s_setprio(2); else if (r4 == 2) __builtin_amdgcn_s_setprio(1);
    }
    extern __shared__ double s_dyn[];                                // weights of pass 2: [TM_TILE_MAXU * TM_UNIT_BINS]   (GRAD only)
    tm_eval_body<GRAD, GEN>(a, chain, tile, s_dyn);
}

int tm_launch_eval(const TmEvalArgs &a, int Nchains, bool grad, void *stream_)
{
    if (a.n_mult > TM_MAXMULT || a.tiles < 1) return (int)hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    dim3 grid(a.tiles, Nchains), block(TM_THREADS);
    if (a.order_mode != 0) grid = dim3(Nchains, a.tiles);
    const size_t lds = grad ? (size_t)TM_TILE_MAXU * TM_UNIT_BINS * sizeof(double) : 8;
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
    const bool gen = a.generic != 0;
    if (grad && gen)       hipLaunchKernelGGL((tamcmc_eval_kernel<true, true>), grid, block, lds, stream, a);
    else if (grad)         hipLaunchKernelGGL((tamcmc_eval_kernel<true, false>), grid, block, lds, stream, a);
    else if (gen)          hipLaunchKernelGGL((tamcmc_eval_kernel<false, true>), grid, block, lds, stream, a);
    else                   hipLaunchKernelGGL((tamcmc_eval_kernel<false, false>), grid, block, lds, stream, a);
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
