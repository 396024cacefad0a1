// tamcmc_fused.hip -- short grids (one tile per chain): the per-chain prologue and the evaluation in ONE launch.
//
// A local fit of the reference's own example has ~1000 bins and ten chains: there the step is launch latency, not
// arithmetic (setup 9.5 us + eval 6 us + the gap, against ~2 us of work).  When the geometry gives every chain a single
// tile, workgroup c runs tm_setup_body for chain c (tamcmc_setup_body.h: same code, same non-contracted arithmetic as
// tamcmc_setup_kernel) and then tm_eval_body for the chain's only tile (tamcmc_eval_body.h: same code as
// tamcmc_eval_kernel), so the results are those of the two-launch path bit for bit.
//
// The eval body reads the records through the constant address space (scalar loads through the scalar data cache,
// which is NOT coherent with vector stores and is shared by neighbouring CUs).  Here the same launch has just written
// them, so
//  (i)   the stores are made visible first (__threadfence: the vector stores reach L2, which the scalar cache misses
//        into);
//  (ii)  the scalar cache is invalidated (s_dcache_inv) before the first record is read.  The per-chain records are not
//        multiples of a cache line (TmNoise 120 B, TmMult 160 B x n_mult, the active list 4 B x n_mult, wt 16 B), so the
//        records of chains c and c+1 share lines: a workgroup that scalar-loads such a line BEFORE its neighbour's
//        stores have landed leaves a stale copy behind, which the neighbour -- if it shares the scalar cache -- would
//        then hit.  (In SPX mode consecutive workgroups go to different XCDs, which hides this; nothing guarantees it.)
//  (iii) the pointers handed to the eval body are passed through an opaque asm, so that the compiler cannot schedule a
//        constant-address-space load -- which it may otherwise move anywhere -- ahead of the barrier.
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#include "tamcmc_setup_body.h"
#include "tamcmc_eval_body.h"

template <bool GRAD>
__global__ __launch_bounds__(TM_THREADS) void tamcmc_fused_kernel(TmLayout L, TmFusedArgs f, TmEvalArgs a, TmCostModel cm)
{
    extern __shared__ double s_dyn[];   // [f.p_doubles] this chain's params row, then [TM_TILE_MAXU * TM_UNIT_BINS] weights (GRAD)
    const int chain = blockIdx.x;
    const int units = (a.Nx + TM_UNIT_BINS - 1) >> TM_UNIT_SHIFT;
    tm_setup_body<TM_THREADS>(L, chain, f.params, f.Tcoefs, const_cast<double *>(a.wt), a.lx, units, a.cells, 1, 0, cm,
                              const_cast<TmMult *>(a.mult), const_cast<TmNoise *>(a.noise), const_cast<TmCellRec *>(a.cell),
                              const_cast<TmTileHdr *>(a.thdr), const_cast<TmActive *>(a.tidx), static_cast<TmChain *>(f.chain_rec),
                              static_cast<TmMultFull *>(f.aux), f.hser, nullptr, s_dyn, nullptr);
    __threadfence();
    __syncthreads();
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    TmEvalArgs b = a;
    asm volatile("" : "+s"(b.mult), "+s"(b.noise), "+s"(b.cell), "+s"(b.thdr), "+s"(b.tidx), "+s"(b.wt) : : "memory");
    tm_eval_body<GRAD>(b, chain, 0, s_dyn + f.p_doubles);
}

int tm_launch_fused(const TmLayout &L, const TmFusedArgs &f, const TmEvalArgs &a, int Nchains, bool grad, void *stream_)
{
    if (a.n_mult > TM_MAXMULT || a.tiles != 1 || tm_units(a.Nx) > TM_TILE_MAXU) return (int)hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    const int units = tm_units(a.Nx);
    const size_t lds = ((size_t)f.p_doubles + (grad ? (size_t)units * TM_UNIT_BINS : 1)) * sizeof(double);
    TmCostModel cm{0, 0, 0, TM_TILE_MAXU};
    if (grad) hipLaunchKernelGGL((tamcmc_fused_kernel<true>), dim3(Nchains), dim3(TM_THREADS), lds, stream, L, f, a, cm);
    else      hipLaunchKernelGGL((tamcmc_fused_kernel<false>), dim3(Nchains), dim3(TM_THREADS), lds, stream, L, f, a, cm);
    return (int)hipGetLastError();
}
