// Cross-workgroup hand-off latency on gfx950: two workgroups bounce a counter through global memory, N round trips.
// One round trip = store -> seen by the other workgroup -> its store -> seen here; the figure printed is HALF of it (one hand-off),
// which is what sits between two dependent MPPI solves (DESIGN.md 4.12 "hand-off", 9 row 3).
//   placement: workgroup i of a launch runs on XCD i % 8, so (0, 8) share an XCD and its L2, (0, 1) do not.
//   mode 0: device-scope atomic store / load (sc1: the library's granules and counters)
//   mode 1: plain store, sc0 load        -- through ONE XCD's L2 (valid only for same-XCD pairs)
//   mode 2: sc0 store, sc0 load
//   mode 3: sc0 sc1 store / load (system scope)
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_pingpong.hip -o tools/ubench_pingpong.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else if (MODE == 2) asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned ld(unsigned *p)
{
    unsigned v;
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ void pingpong(unsigned *flags, int a, int b, int n, unsigned long long *cycles, unsigned *xcc)
{
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != a && me != b) return;
    unsigned *mine = flags + (me == a ? 0 : 64), *other = flags + (me == a ? 64 : 0);     // 256 bytes apart
    xcc[me == a ? 0 : 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID
    const unsigned long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (me == a) {
            st<MODE>(mine, (unsigned)i);
            long guard = 0;
            while (ld<MODE>(other) != (unsigned)i && ++guard < (1L << 16)) { }
            if (guard >= (1L << 16)) { *cycles = 0; return; }           // never seen (stale line in this XCD's L2): give up, report 0
        } else {
            long guard = 0;
            while (ld<MODE>(other) != (unsigned)i && ++guard < (1L << 17)) { }
            if (guard >= (1L << 17)) return;
            st<MODE>(mine, (unsigned)i);
        }
    }
    if (me == a) *cycles = wall_clock64() - t0;       // 100 MHz
}

template <int MODE>
void run(const char *what, unsigned *flags, unsigned long long *cyc, unsigned *xcc, int a, int b)
{
    const int n = 2000;
    hipMemset(flags, 0, 1024);
    pingpong<MODE><<<16, 64>>>(flags, a, b, n, cyc, xcc);
    hipDeviceSynchronize();
    unsigned long long c; unsigned x[2];
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
    printf("%-44s workgroups (%d,%2d) on XCC (%u,%u): %6.0f ns per hand-off\n", what, a, b, x[0], x[1], (double)c * 10.0 / n / 2.0);
}

int main()
{
    unsigned *flags, *xcc; unsigned long long *cyc;
    hipMalloc(&flags, 1024); hipMalloc(&cyc, 8); hipMalloc(&xcc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("device-scope store / load (sc1)", flags, cyc, xcc, 0, 1);
        run<0>("device-scope store / load (sc1)", flags, cyc, xcc, 0, 8);
        run<3>("system-scope store / load (sc0 sc1)", flags, cyc, xcc, 0, 1);
        run<3>("system-scope store / load (sc0 sc1)", flags, cyc, xcc, 0, 8);
        run<1>("plain store, sc0 load (one XCD's L2)", flags, cyc, xcc, 0, 8);
        run<2>("sc0 store, sc0 load (one XCD's L2)", flags, cyc, xcc, 0, 8);
        run<1>("plain store, sc0 load ACROSS XCDs (unsafe)", flags, cyc, xcc, 0, 1);
    }
    return 0;
}
