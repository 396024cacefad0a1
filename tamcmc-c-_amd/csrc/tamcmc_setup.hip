// tamcmc_setup.hip -- per-chain prologue kernel: params row -> multiplet table, noise record, cell polynomials,
// equal-cost tile boundaries, active-multiplet lists and launch ranks.  Compiled with -ffp-contract=off (see tamcmc_derive.h); the body lives in
// tamcmc_setup_body.h, which the fused small-grid kernel (tamcmc_fused.hip) shares.
//
// One workgroup of eight waves per chain.  Three work side by side on the params row: wave 0 (lane 0 the chain's
// scalars, then a lane per multiplet) the multiplet records, wave 1 the noise record and the per-cell background
// polynomials, wave 2 the inclination and the m-ratio tables.  The truncation windows do not depend on the ratios, so
// they are published first and waves 1..7 build the tiles' active lists while wave 0 completes and stores the records;
// launch ranks last.  The work is O(Nparams) per chain -- a chain of single-wave stages (time line: TM_SU_TRACE,
// tools/su_trace.py); it exists as a kernel so that a sampler can keep params resident in HBM and chain the whole
// evaluation on one stream without a host round trip.
#include <hip/hip_runtime.h>
#include "tamcmc_dev.h"
#include "tamcmc_setup_body.h"

#ifndef TM_SETUP_THREADS
#define TM_SETUP_THREADS 512
#endif
__global__ __launch_bounds__(TM_SETUP_THREADS) void tamcmc_setup_kernel(TmLayout L, const double *__restrict__ params,
                                                          const double *__restrict__ Tcoefs, double *__restrict__ wt,
                                                          const double *__restrict__ lx, int units, int cells, int tiles, int equal_cost,
                                                          TmCostModel cm, int p_doubles,
                                                          TmMult *__restrict__ mult, TmNoise *__restrict__ noise,
                                                          TmCellRec *__restrict__ cell, TmTileHdr *__restrict__ thdr, TmActive *__restrict__ tidx,
                                                          TmChain *__restrict__ chain_rec, TmMultFull *__restrict__ aux,
                                                          double *__restrict__ hser, int32_t *__restrict__ order)
{
    extern __shared__ double s_p[];   // this chain's params row (every later access is an LDS read), then the unit-cost prefix
    int *s_pre = (equal_cost != 0) ? reinterpret_cast<int *>(s_p + p_doubles) : nullptr;
    tm_setup_body<TM_SETUP_THREADS>(L, (int)blockIdx.x, params, Tcoefs, wt, lx, units, cells, tiles, equal_cost, cm, mult, noise, cell, thdr,
                                    tidx, chain_rec, aux, hser, order, s_p, s_pre);
}

int tm_launch_setup(const TmLayout &L, int Nchains, const double *d_params, const double *d_Tcoefs, double *d_wt, const double *d_lx,
                    int units, int cells, int tiles, int equal_cost, TmCostModel cm, TmMult *d_mult, TmNoise *d_noise, TmCellRec *d_cell,
                    TmTileHdr *d_thdr, TmActive *d_tidx, void *d_chain_rec, void *d_aux, double *d_hser, int32_t *d_order, void *stream)
{
    if (tiles < 1 || cm.pad < 1 || (long long)tiles * cm.pad < units) return (int)hipErrorInvalidValue;   // cm.pad: units per tile at most
    const int p_doubles = (L.Nparams + 1) & ~1;
    const int eq = tm_setup_balances(units, tiles, equal_cost, cm.pad);
    const size_t lds = (size_t)p_doubles * sizeof(double) + (eq ? (size_t)units * sizeof(int) : 0);
    hipLaunchKernelGGL(tamcmc_setup_kernel, dim3(Nchains), dim3(TM_SETUP_THREADS), lds, (hipStream_t)stream, L,
                       d_params, d_Tcoefs, d_wt, d_lx, units, cells, tiles, eq, cm, p_doubles, d_mult, d_noise, d_cell, d_thdr, d_tidx,
                       static_cast<TmChain *>(d_chain_rec), static_cast<TmMultFull *>(d_aux), d_hser, d_order);
    return (int)hipGetLastError();
}

size_t tm_sizeof_chain_rec() { return sizeof(TmChain); }
size_t tm_sizeof_aux() { return sizeof(TmMultFull); }

// Gate of an armed batch (tamcmc_eval_batch_arm / _fire, tamcmc_api.cpp): one wave that holds the stream until the host
// has put the batch's parameters into the pinned input buffer and stored `target` into the pinned word gate[0].  The
// launches of the batch sit behind it in the stream, so what the host pays between knowing the parameters and the GPU
// starting on them is one store, not two kernel launches.
// A wave must never outlive its host, so the wait is bounded (`patience` polls, ~2 us each: ~4 s by default).  When it
// runs out the gate says so in gate[TM_GATE_EXPIRED] and lets the batch run on whatever the input buffer holds; the
// host looks at that word when it fires and, if the gate has given up, throws the stale batch away and launches again
// (a host that was merely slow -- a debugger, a device-wide synchronisation somewhere else in the process -- loses
// time, never a result).  The two sides may meet: the host announces itself in gate[TM_GATE_FIRING] before it looks at
// the expiry word, the gate looks at that announcement after it has written the expiry word (store-then-load on both
// sides: at least one sees the other); a gate that finds the host firing goes on waiting for the word to open.
#define TM_GATE_EXPIRED 16     // (uint32 index: one cache line apart)
#define TM_GATE_FIRING 32
__global__ __launch_bounds__(64) void tamcmc_gate_kernel(uint32_t *gate, uint32_t target, int patience)
{
    if (threadIdx.x != 0) return;
    auto open = [&]() { return (int32_t)(__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) >= 0; };
    for (int spin = 0; spin < patience; spin++) {
        if (open()) return;
        __builtin_amdgcn_s_sleep(16);
    }
    __hip_atomic_store(gate + TM_GATE_EXPIRED, target, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__hip_atomic_load(gate + TM_GATE_FIRING, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM) != target) return;   // expired
    for (int spin = 0; spin < (1 << 20); spin++) {      // the host is firing right now: the word opens within microseconds
        if (open()) return;
        __builtin_amdgcn_s_sleep(16);
    }
}

int tm_launch_gate(uint32_t *dv_gate, uint32_t target, int patience, void *stream)
{
    hipLaunchKernelGGL(tamcmc_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dv_gate, target, patience);
    return (int)hipGetLastError();
}
