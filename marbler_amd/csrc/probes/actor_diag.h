// probes/actor_diag.h -- the diagnostic hooks of csrc/actor_mfma.hip.  In the shipped build every macro here is empty; with
// -DRG_ACTOR_STAMPS (tools/actor_stamps.py, tools/actor_lab) a wave stamps s_memtime at its phase boundaries and writes the stamps
// behind the q block; -DRG_ACTOR_STAMPS_FC1 stamps the head of the wave in detail instead (slots 2..5).
#pragma once

#ifdef RG_ACTOR_STAMPS
#define RG_ASTAMP_BEGIN()                                                                                             \
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), rt_start = __builtin_amdgcn_s_memrealtime();     \
    int stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RG_ASTAMP(i) stamps[i] = static_cast<int>(__builtin_amdgcn_s_memtime() - t_start)
// keep values alive up to the stamp (the stamp must not float above the work it closes)
#define RG_AKEEP1(x) asm volatile("" ::"v"(x))
#define RG_AKEEP2(x, y) asm volatile("" ::"v"(x), "v"(y))
#ifdef RG_ACTOR_STAMPS_FC1   // the head of the wave in detail (slots 2..5; the later phases' stamps are left out)
#define RG_HSTAMP(i) RG_ASTAMP(i)
#define RG_HKEEP2(x, y) asm volatile("" ::"v"(x), "v"(y))
#define RG_PSTAMP(i)
#else
#define RG_HSTAMP(i)
#define RG_HKEEP2(x, y)
#define RG_PSTAMP(i) RG_ASTAMP(i)
#endif
// slot 6: the end; slot 7: the wave's life on the constant 100 MHz clock (slot 6 / slot 7 = shader clock / 100 MHz); slot 0: where
// the wave ran -- XCC_ID (hwreg 20) << 16 | HW_ID (hwreg 4: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
#define RG_ASTAMP_END(a, E, N, A, H, cb, lane)                                                                                    \
    RG_ASTAMP(6);                                                                                                                  \
    stamps[7] = static_cast<int>(__builtin_amdgcn_s_memrealtime() - rt_start);                                                     \
    stamps[0] = static_cast<int>((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 16) |                                    \
                                 (__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4) & 0xFFFF));                                 \
    __syncthreads();                                                                                                               \
    if ((lane) == 0 && (a).q) {                                                                                                    \
        int *dst = reinterpret_cast<int *>((a).q) + static_cast<size_t>(E) * (N) * (A) +                                           \
                   (static_cast<size_t>(blockIdx.x) * ((H) / 32) + (cb)) * 8; /* behind the q block */                             \
        for (int i = 0; i < 8; ++i) dst[i] = stamps[i];                                                                            \
    }
// what the runtime says about co-resident workgroups per CU
#define RG_ACTOR_DIAG_ENTRY                                                                                           \
    extern "C" int rg_actor_occupancy(int hidden_dim) {                                                               \
        int n = -1;                                                                                                   \
        if (hidden_dim == 64) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rg::actor_kernel<64, 1>, 128, 0); \
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rg::actor_kernel<128, 1>, 256, 0);                \
        return n;                                                                                                     \
    }
#else
#define RG_ASTAMP_BEGIN()
#define RG_ASTAMP(i)
#define RG_AKEEP1(x)
#define RG_AKEEP2(x, y)
#define RG_HSTAMP(i)
#define RG_HKEEP2(x, y)
#define RG_PSTAMP(i)
#define RG_ASTAMP_END(a, E, N, A, H, cb, lane)
#define RG_ACTOR_DIAG_ENTRY
#endif
