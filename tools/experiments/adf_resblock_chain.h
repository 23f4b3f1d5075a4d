// EXPERIMENT, not shipped (round 4; VERDICT r3 item 3 "one persistent launch for the sample-local bottom of the U"): the short-level resblocks of a level chained in
// ONE launch of the same B x 4 workgroups as adf_resblock_split.h, with a barrier among the four workgroups of a sample between the phases and between the blocks.
// It was built into the product, tested and measured, then taken out:
//   * with agent-scope FENCES around the barrier (release = buffer_wbl2 sc1, acquire = buffer_inv sc1: the XCD's whole L2 is invalidated at every barrier of every
//     workgroup) the chained kernels ran 2.7 x SLOWER than the launches they replace: 304.2 / 303.8 / 304.1 against 231.5 / 230.8 / 234.0 ms per bench step;
//   * with agent-scope ("sc1") loads / stores for the exchanged rows and statistics only, no fence (below): 238.7 / 239.2 / 239.3 against 237.2 / 236.9 / 236.4 ms per
//     step (profiles/r04_ab_resblock_chain.txt) -- still 1 % slower: workgroups are dealt to the 8 XCDs round-robin, so the four siblings of a sample sit on four XCDs
//     and an arrival (device-scope atomic to the memory side + polling loads) costs what a launch boundary inside a captured graph costs, ~2-3 us; and the
//     bit-for-bit route test against the two-launch form, green in the A/B calls, failed once in the full suite (class-conditional variant): `s_waitcnt vmcnt(0)`
//     behind write-through stores is evidently not the release the exchange needs -- the correct one is the buffer_wbl2 of the first variant.
// Kept as the record of why the per-launch latency of the bottom of the U (0.44 ms of a 2.33 ms evaluation in 33 launches) is not recovered by in-kernel barriers on a
// part whose L2s are per XCD.  To try it again: paste the two blocks below back into adf_resblock_split.h (the load / run split of the kernel body they rely on is in
// the product) and the flush_chain() of git revision 5d53a5e..HEAD~ into the walker.
#pragma once
#include "../../audiodiffuser_amd/csrc/adf_resblock_split.h"

namespace adf {

// Agent-scope ("sc1") accesses for data that workgroups on different XCDs exchange INSIDE one launch (adf_resblock_split.h, chained kernel): each XCD has its
// own L2, which is only written back / invalidated at kernel boundaries; a relaxed atomic at agent scope goes to the level all XCDs share.  No fence is used
// for that exchange: an agent-scope acquire invalidates the XCD's whole L2 -- measured, the chained kernel with fences ran 2.7 x SLOWER than the launches it
// replaced (every weight stream then came from memory).
__device__ __forceinline__ u32x4_t coh_load16(const void* p) {
    const unsigned long long* q = (const unsigned long long*)p;
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return u32x4_t{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
__device__ __forceinline__ void coh_store16(void* p, const u32x4_t& v) {
    unsigned long long* q = (unsigned long long*)p;
    __hip_atomic_store(q, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coh_load_f64(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void coh_store_f64(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- the blocks of one level chained in ONE launch (round 4) ----------------------------------------------------------------------------------------
// Up to three consecutive resblocks of a level (down: 2, bottleneck: 1, up: 3) as one launch of the same B x 4 workgroups: the exchange between a block's
// two phases and between consecutive blocks -- everything a launch boundary was needed for -- is a barrier among the FOUR workgroups of a sample: each
// arrives on a per-sample counter in global memory and spins until its three siblings have.  The next phase's weight fragments are requested BEFORE the
// barrier (they do not depend on the exchange), so their L2 round trip overlaps it.  A spin on sibling workgroups hangs if they are not co-resident: the
// launcher takes this route only when hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs covers the whole grid (else the two-launch form above), and the
// spin gives up after ~0.3 s, counting a fault (adf_run_counters::device_faults) instead of hanging the queue.
constexpr int kRbChainMax = 3;
struct RbChainArgs {
    int nblk;
    int cin[kRbChainMax];
    RbSplitArgs blk[kRbChainMax];
    unsigned* flags;          // [B] arrival counters, zero at launch (the statistics arena: one memset per network pass)
    unsigned* faults;         // device counter of barriers that timed out
};

__device__ __forceinline__ void rb_group_barrier(unsigned* flag, unsigned target, unsigned* faults) {
    // every wave: its (agent-scope) stores of this phase have been acknowledged before the arrival.  No fence: the exchanged tensors are written and read
    // with agent-scope accesses (COH), everything else -- weights, parameters -- stays in this XCD's L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21)) { __hip_atomic_fetch_add(faults, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

template <int NTOK, int CIN>
__device__ __forceinline__ void rb_chain_block(const RbSplitArgs& aa, char* smem, int bq, bool first, unsigned* flag, unsigned& arrivals, unsigned* faults) {
    {
        RbSplitWf<NTOK, CIN, 1> W;
        rb_split_load_w<NTOK, CIN, 1>(aa, bq, W);
        if (!first) { arrivals += 4; rb_group_barrier(flag, arrivals, faults); }       // the previous block's output rows and statistics
        rb_split_run<NTOK, CIN, 1, true>(aa, smem, bq, W);
    }
    {
        RbSplitWf<NTOK, CIN, 2> W;
        rb_split_load_w<NTOK, CIN, 2>(aa, bq, W);
        arrivals += 4;
        rb_group_barrier(flag, arrivals, faults);                                       // all four column quarters of silu(FiLM(GN2(h1)))
        rb_split_run<NTOK, CIN, 2, true>(aa, smem, bq, W);
    }
}

template <int NTOK>
__global__ void __launch_bounds__(512) resblock_chain_kernel(const RbChainArgs c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bq = (int)blockIdx.x;
    unsigned* const flag = c.flags + (bq >> 2);
    unsigned arrivals = 0;
#define ADF_RB_CHAIN_BLOCK(k)                                                                                              \
    if (k < c.nblk) {                                                                                                      \
        if (c.cin[k] == 256) rb_chain_block<NTOK, 256>(c.blk[k], smem, bq, k == 0, flag, arrivals, c.faults);              \
        else rb_chain_block<NTOK, 512>(c.blk[k], smem, bq, k == 0, flag, arrivals, c.faults);                              \
    }
    ADF_RB_CHAIN_BLOCK(0)
    ADF_RB_CHAIN_BLOCK(1)
    ADF_RB_CHAIN_BLOCK(2)
#undef ADF_RB_CHAIN_BLOCK
}


// 0 = launched; 1 = not taken (the grid would not be co-resident on this device: use the two-launch form); else *err is set
template <int NTOK>
inline int launch_resblock_chain_t(const RbChainArgs& c, int B, hipStream_t s, const char** err) {
    static int resident_dev[kMaxDevices] = {};       // 0 = not asked yet, -1 = attribute / query failed, else workgroups that can be resident at once
    int& resident = resident_dev[current_device()];
    const size_t lds = resblock_split_lds(NTOK, 512);
    if (resident == 0) {
        resident = -1;
        int per_cu = 0, cus = 0, dev = 0;
        if (hipFuncSetAttribute((const void*)resblock_chain_kernel<NTOK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)resblock_chain_kernel<NTOK>, 512, lds) == hipSuccess &&
            hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
            resident = per_cu * cus;
    }
    if (resident < B * 4 || lds > 160 * 1024) return 1;
    hipLaunchKernelGGL((resblock_chain_kernel<NTOK>), dim3(B * 4), dim3(512), lds, s, c);
    if (hipGetLastError() != hipSuccess) { *err = "resblock_chain: launch failed"; return 2; }
    return 0;
}
inline int launch_resblock_chain(const RbChainArgs& c, int B, int ntok, hipStream_t s, const char** err) {
    *err = nullptr;
    if (c.nblk < 1 || c.nblk > kRbChainMax || !c.flags || !c.faults) return 1;
    for (int k = 0; k < c.nblk; ++k) if (c.cin[k] != 256 && c.cin[k] != 512) return 1;
    if (ntok == 64) return launch_resblock_chain_t<64>(c, B, s, err);
    if (ntok == 16) return launch_resblock_chain_t<16>(c, B, s, err);
    return 1;
}


}  // namespace adf
