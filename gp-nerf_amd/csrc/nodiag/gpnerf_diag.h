// gpnerf_diag.h, PRODUCT version (csrc/nodiag/, the include path of `make libgpnerf_hip.so`).
//
// The kernels carry a handful of named hook points for the lab (cycle stamps per phase, per-wavefront time stamps, launcher
// experiment knobs read from the environment).  In the product every hook is empty, resolved at compile time: no
// diagnostic code, no getenv, no extra export exists in libgpnerf_hip.so, and no -D switch can change that -- the lab's
// versions of these definitions live in csrc/diag/gpnerf_diag.h and are reached only through csrc/diag/Makefile (-I order),
// which builds differently named libraries that tools/ load via GPNERF_DEBUG=1 GPNERF_LIB_PATH=...
// Included once per translation unit, inside its anonymous namespace, after DEV is defined.
#pragma once
#ifndef DEV
#define DEV __device__ __forceinline__
#endif
struct Stamps { DEV void start() {} DEV void flush(int) {} };
#define STAMP(st, i) ((void)0)
#define STAMP_T0() ((void)0)
#define STAMP_ADD(i, lane) ((void)0)
#define WT(i) ((void)0)
#define WT_COUNT() ((void)0)
// launcher knobs: the product always takes the default
inline const char* dbg_env(const char*) { return nullptr; }
inline int dbg_int(const char*, int dflt, int, int) { return dflt; }
// extern "C" block of gpnerf_kernels.hip: nothing beyond include/gpnerf_hip.h
#define GPNERF_DIAG_EXPORTS
