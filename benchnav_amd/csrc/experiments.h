// experiments.h -- the measurement builds' hooks into the kernels; included by mppi_device.h under -DBN_EXPERIMENTS only
// (tools/build_variant*.py).  The shipped library (benchnav_amd/build.py) never sees this file.
//   BN_TIMING          in-kernel cycle / wall-clock stamps of workgroup 0 (tools/stamps*.py) and per-workgroup traces
//                      (tools/block_trace*.py): SolveParams::stamps must point at a device buffer (bn_mppi_debug_set_stamps)
//   BN_WAVE_LDS_PAD=n  the one-wave kernel asks for n more bytes of LDS: fewer workgroups per CU (tools/wave_ab.py, the occupancy
//                      sweep of profiles/r5_experiments/occupancy.txt); read by wave_lds_bytes in mppi_kernels.hip
#pragma once
#ifdef BN_TIMING
#define BN_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { /* instance 0, workgroup 0 */ \
            p.stamps[slot] = __builtin_readcyclecounter();                                               \
            if ((slot) < 16) p.stamps[32 + ((p.solve & 1) << 4) + (slot)] = wall_clock64();   /* chip-wide 100 MHz clock, by solve parity (tools/stamps_overlap.py) */ \
        }                                                                                                \
    } while (0)
#define BN_STAMP_ANY(slot)                                                                               \
    do {                                                                                                 \
        if (p.stamps && blockIdx.y == 0 && threadIdx.x == 0) p.stamps[slot] = __builtin_readcyclecounter(); \
    } while (0)
// per-wave cycle stamps of workgroup 0 (tools/stamps_overlap.py): slot i of wave w at stamps[192 + 12 w + i]
#define BN_WSTAMP(i)                                                                                     \
    do {                                                                                                 \
        if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)                   \
            p.stamps[192 + 12 * (threadIdx.x >> 6) + (i)] = __builtin_readcyclecounter();                \
    } while (0)
// per-workgroup trace (tools/block_trace.py): wall clock (100 MHz, chip-wide) at entry and exit, cycles, HW_ID
#define BN_TRACE_BEGIN()                                                                                 \
    const unsigned long long bn_tr_t0 = wall_clock64(), bn_tr_c0 = __builtin_readcyclecounter()
#define BN_TRACE_END()                                                                                   \
    do {                                                                                                 \
        if (p.stamps && threadIdx.x == 0) {                                                              \
            unsigned long long *r = p.stamps + 64 + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x + (p.trace_by_parity ? (size_t)(p.solve & 1) * gridDim.x * gridDim.y : 0));   \
            r[0] = bn_tr_t0; r[1] = wall_clock64(); r[2] = __builtin_readcyclecounter() - bn_tr_c0;      \
            r[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
        }                                                                                                \
    } while (0)
#define BN_TIMING_DO(...) __VA_ARGS__
// the self tail of a one-launch forward (finish_body<.., SELF>): chip-wide 100 MHz clock of instance 0's tail workgroup at stamps[900 + i]
// (tools/stamps_forward.py); the rollout workgroup (0, 0) of the same launch leaves its own at stamps[32 + 16 parity + slot]
#define BN_SSTAMP(i) do { if (SELF && p.stamps && threadIdx.x == 0 && b == 0) p.stamps[900 + 16 * (p.tail_solve & 1) + (i)] = wall_clock64(); } while (0)
#else
#define BN_SSTAMP(i) do { } while (0)
#define BN_TIMING_DO(...)
#define BN_STAMP(slot) do { } while (0)
#define BN_STAMP_ANY(slot) do { } while (0)
#define BN_WSTAMP(i) do { } while (0)
#define BN_TRACE_BEGIN() do { } while (0)
#define BN_TRACE_END() do { } while (0)
#endif
