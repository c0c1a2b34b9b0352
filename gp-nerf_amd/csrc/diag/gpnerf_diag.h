// gpnerf_diag.h, LAB version (csrc/diag/): what the hook points of the kernels expand to in the diagnostic libraries that
// csrc/diag/Makefile builds (libgpnerf_hip_diag.so, _stamps.so, _wavetimes.so).  Never on the product's include path; the
// product's (empty) version is csrc/nodiag/gpnerf_diag.h.  tools/ load these libraries with GPNERF_DEBUG=1 GPNERF_LIB_PATH=...
//   -DGPNERF_STAMPS      per-phase cycle shares of the fused kernel (tools/stamps.py)
//   -DGPNERF_WAVETIMES   per-wavefront entry / staged / first-step / exit times (tools/wave_times.py)
//   (always)             launcher experiment knobs from the environment, clamped: GPNERF_WAVES, GPNERF_SPLIT, GPNERF_QSPLIT,
//                        GPNERF_CHAIN_*, GPNERF_QUEUE_CHUNK, GPNERF_CONV_*, GPNERF_ATT_* ... (tools/probes/*.sh)
#pragma once
#ifndef DEV
#define DEV __device__ __forceinline__
#endif
#ifdef GPNERF_STAMPS
__device__ unsigned long long g_stamps[16];
struct Stamps {
    unsigned long long prev, acc[16];
    DEV void start() {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        prev = now();
    }
    DEV static unsigned long long now() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    DEV void mark(int i) { const unsigned long long t = now(); acc[i] += t - prev; prev = t; }
    DEV void flush(int lane) {
        if (lane == 0)
            for (int i = 0; i < 16; ++i) atomicAdd(&g_stamps[i], acc[i]);
    }
};
#define STAMP(st, i) (st).mark(i)
#define STAMP_T0() const unsigned long long stamp_t0 = Stamps::now()
#define STAMP_ADD(i, lane) do { if ((lane) == 0) { atomicAdd(&g_stamps[i], Stamps::now() - stamp_t0); atomicAdd(&g_stamps[(i) + 4], 1ull); } } while (0)
#else
struct Stamps { DEV void start() {} DEV void flush(int) {} };
#define STAMP(st, i) ((void)0)
#define STAMP_T0() ((void)0)
#define STAMP_ADD(i, lane) ((void)0)
#endif
// when each wavefront enters the kernel, has its weights, ends its first sample step and leaves (100 MHz real-time counter,
// comparable across the chip)
#ifdef GPNERF_WAVETIMES
__device__ unsigned long long g_wt[16384 * 4];
#define WT(i) do { if ((threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    g_wt[((blockIdx.x * 8u + (threadIdx.x >> 6)) & 16383u) * 4u + (i)] = t_; } } while (0)
#define WT_COUNT() do { if ((threadIdx.x & 63) == 0) g_wt[((blockIdx.x * 8u + (threadIdx.x >> 6)) & 16383u) * 4u + 2] += 1ull << 48; } while (0)
#else
#define WT(i) ((void)0)
#define WT_COUNT() ((void)0)
#endif
// Experiment knobs (tools/*.sh A/B runs): read ONLY when GPNERF_DEBUG=1 is set, and clamped to [lo, hi]
inline const char* dbg_env(const char* name) {
    static int on = -1;
    if (on < 0) { const char* d = getenv("GPNERF_DEBUG"); on = (d && d[0] == '1') ? 1 : 0; }
    return on ? getenv(name) : nullptr;
}
inline int dbg_int(const char* name, int dflt, int lo, int hi) {
    const char* e = dbg_env(name);
    if (!e) return dflt;
    const int v = atoi(e);
    return v < lo ? lo : (v > hi ? hi : v);
}
// the diagnostic libraries' extra exports (expanded inside gpnerf_kernels.hip's extern "C" block, after the anonymous namespace)
#if defined(GPNERF_WAVETIMES) && defined(GPNERF_STAMPS)
#error "one diagnostic at a time"
#endif
#if defined(GPNERF_WAVETIMES)
#define GPNERF_DIAG_EXPORTS \
    int gpnerf_debug_read_wavetimes(unsigned long long* out, int n_waves) { \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wt), sizeof(unsigned long long) * 4 * (size_t)n_waves) == hipSuccess ? GPNERF_OK : GPNERF_E_DEVICE; \
    }
#elif defined(GPNERF_STAMPS)
#define GPNERF_DIAG_EXPORTS \
    int gpnerf_debug_read_stamps(unsigned long long* out16) { \
        if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return GPNERF_E_DEVICE; \
        unsigned long long zero[16] = {0}; \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero)) == hipSuccess ? GPNERF_OK : GPNERF_E_DEVICE; \
    }
#else
#define GPNERF_DIAG_EXPORTS
#endif
