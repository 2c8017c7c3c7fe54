// What does a waiting launch's poll cost while its predecessor publishes?  (DESIGN_NOTEBOOK.md R5.4: in the product the polls that
// overlap the publication take 0.6-1.1 us instead of 0.41, and the period of dependent solves moves in steps of one such poll.)
// 16 writer workgroups publish one row of 102 granules ({value, tag}, 8 bytes) each, every 12 us by the chip-wide 100 MHz clock, the way
// the latency kernel does it (five waves, ~7 wave-instructions of 8-byte device-scope stores spread over ~0.5 us, one more for the two
// leading granules) or in one burst (51 lanes x 16 bytes); 16 poller workgroups poll all 16 rows back to back the way its fast prologue
// does (three waves: 8 columns / 64 columns / the rest, 16 + 2 loads per lane) and stamp the clock when every granule carries the
// generation's tag.  Printed: how long after the LAST writer's last store instruction the pollers had the rows, and how long the
// polls around that moment took.  One workgroup per CU (80 KB of LDS each), 200 generations.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_publish.hip -o tools/ubench_publish.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int kRows = 16, kCols = 100, kPS = 2 + kCols, kGen = 200;
constexpr unsigned long long kPeriod = 1200;             // 12 us in 10 ns ticks

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }
__device__ __forceinline__ void st_gran(unsigned long long *p, float v, uint32_t tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE bit 0: burst publication (one wave, 16-byte stores) instead of the scattered pattern; bit 1: only the chain-like wave polls
// (8 columns + the leading pair) until it has seen them, the other two waves start then; bit 2: pollers sleep ~0.1 us between polls;
// bit 3: only two of the sixteen waiting workgroups poll; bit 4: waves 2 and 3 poll ONE granule per row (its last column, written in the
// second pass) plus the leading pair, and read their columns once behind it; bit 5: the pollers poll ANOTHER array that nobody writes
// (same size, same pattern) and give up 3 us after the publication: is a poll slow because of the stores as such, or because they go to
// the lines it reads?  bit 6: one SENTINEL per row in a line of its own (256 bytes apart), written behind the row's last store without
// waiting for anything; wave 1 polls the sixteen sentinels (one load in sixteen lanes), then all three waves read their granules, which
// validate themselves by their tags (read again until they do).  (`seen` then holds the give-up time; the poll durations printed are the longest of the window and its neighbours)
template <int MODE>
__global__ __launch_bounds__(320) void k_pub(unsigned long long *gran, unsigned long long t0, unsigned long long *wlast, unsigned long long *seen,
                                             unsigned long long *polls)
{
    extern __shared__ float smem[];
    const int wg = blockIdx.x, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    if (tid == 0) smem[0] = 0.0f;
    __shared__ int flag;
    if (wg < kRows) {
        unsigned long long *row = gran + (size_t)wg * kPS;
        for (int g = 0; g < kGen; ++g) {
            const unsigned long long tg = t0 + (unsigned long long)(g + 1) * kPeriod;
            const uint32_t tag = (uint32_t)g + 1u;
            if (MODE & 1) {
                if (wid == 1) {
                    while (wall() < tg + 50) { }
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    if (lane < kPS / 2) {
                        const u4 v = {__float_as_uint(1.0f + lane), tag, __float_as_uint(2.0f + lane), tag};
                        unsigned long long *dst = row + 2 * lane;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
                    }
                    if (lane == 0) wlast[(size_t)g * kRows + wg] = wall();
                    if ((MODE & 64) && lane == 0) st_gran(gran + 8192 + 32 * wg, 0.0f, tag);
                }
            } else {
                // pass 1: 320 items, item = thread: column j = tid >> 2, lanes with (tid & 3) == 0 store; waves start 0-0.1 us apart
                while (wall() < tg + 10 * wid) { }
                { const int j = tid >> 2; if ((tid & 3) == 0 && j < kCols) st_gran(row + 2 + j, 1.0f + j, tag); }
                // pass 2 (threads 0 .. 79) ~0.27 us later; the leading pair by thread 192 ~0.05 us after its pass 1
                if (wid <= 1) {
                    while (wall() < tg + 27 + 10 * wid) { }
                    const int j = (tid + 320) >> 2;
                    if ((tid & 3) == 0 && j < kCols) st_gran(row + 2 + j, 1.0f + j, tag);
                    if (tid == 0) wlast[(size_t)g * kRows + wg] = wall();
                    if ((MODE & 64) && tid == 0) st_gran(gran + 8192 + 32 * wg, 0.0f, tag);      // the sentinel: behind this wave's last store
                }
                if (tid == 192) { while (wall() < tg + 35) { } st_gran(row, 0.5f, tag); st_gran(row + 1, 0.25f, tag); }
            }
        }
        return;
    }
    // pollers: waves 1 .. 3 as in the fast prologue (wave 1: columns lane & 7; wave 2: columns 0 .. 63; wave 3: 64 .. 99)
    if (wid < 1 || wid > 3) return;
    const int p_ = wg - kRows;
    if ((MODE & 8) && p_ >= 2) return;
    const int col = wid == 1 ? (lane & 7) : wid == 2 ? lane : min(64 + lane, kCols - 1);
    if (tid == 64) flag = 0;
    for (int g = 0; g < kGen; ++g) {
        const unsigned long long tg = t0 + (unsigned long long)(g + 1) * kPeriod;
        const uint32_t tag = (uint32_t)g + 1u;
        while (wall() < tg - 300) { }                       // a launch polls for ~3 us before its rows come
        if ((MODE & 2) && wid != 1) { while (*(volatile int *)&flag != g + 1) __builtin_amdgcn_s_sleep(1); }
        bool got = false;
        unsigned long long ts = wall(), te = ts, prev = 0, prev2 = 0;
        if (MODE & 64) {
            if (wid == 1) {
                bool all = false;
                while (!all) {
                    ts = wall();
                    const unsigned long long sv = ld_gran(gran + 8192 + 32 * min(lane, kRows - 1));
                    all = __builtin_amdgcn_ballot_w64((uint32_t)(sv >> 32) != tag) == 0;
                    if (!all) { prev2 = prev; prev = wall() - ts; }
                }
                if (lane == 0) *(volatile int *)&flag = g + 1;
            } else {
                while (*(volatile int *)&flag != g + 1) __builtin_amdgcn_s_sleep(1);
            }
            while (!got) {
                ts = wall();
                unsigned long long v[kRows], mi, si;
#pragma unroll
                for (int i = 0; i < kRows; ++i) v[i] = ld_gran(gran + (size_t)i * kPS + 2 + col);
                mi = ld_gran(gran + (size_t)min(lane, kRows - 1) * kPS);
                si = ld_gran(gran + (size_t)min(lane, kRows - 1) * kPS + 1);
                bool ok = (uint32_t)(mi >> 32) == tag && (uint32_t)(si >> 32) == tag;
#pragma unroll
                for (int i = 0; i < kRows; ++i) ok = ok && (uint32_t)(v[i] >> 32) == tag;
                got = __builtin_amdgcn_ballot_w64(!ok) == 0;
                te = wall();
            }
            if (lane == 0) {
                const size_t o = ((size_t)g * kRows + p_) * 3 + (wid - 1);
                seen[o] = te; polls[o * 3 + 0] = te - ts; polls[o * 3 + 1] = prev; polls[o * 3 + 2] = prev2;
            }
            continue;
        }
        if (MODE & 32) {
            const unsigned long long *other = gran + 4096;   // (32 KB behind the rows: allocated, never written)
            unsigned long long worst = 0, first = 0;
            while (wall() < tg + 300) {
                ts = wall();
                unsigned long long acc = 0;
#pragma unroll
                for (int i = 0; i < kRows; ++i) acc += ld_gran(other + (size_t)i * kPS + 2 + col);
                acc += ld_gran(other + (size_t)min(lane, kRows - 1) * kPS) + ld_gran(other + (size_t)min(lane, kRows - 1) * kPS + 1);
                asm volatile("" :: "v"(acc));
                te = wall();
                if (ts < tg && first == 0) first = te - ts;
                if (ts >= tg && ts < tg + 150) worst = max(worst, te - ts);
            }
            if (lane == 0) {
                const size_t o = ((size_t)g * kRows + p_) * 3 + (wid - 1);
                seen[o] = te; polls[o * 3 + 0] = worst; polls[o * 3 + 1] = first; polls[o * 3 + 2] = 0;
            }
            continue;
        }
        while (!got) {
            ts = wall();
            unsigned long long v[kRows], mi, si;
#pragma unroll
            for (int i = 0; i < kRows; ++i) v[i] = ld_gran(gran + (size_t)i * kPS + 2 + (((MODE & 16) && wid != 1) ? kCols - 1 : col));
            mi = ld_gran(gran + (size_t)min(lane, kRows - 1) * kPS);
            si = ld_gran(gran + (size_t)min(lane, kRows - 1) * kPS + 1);
            bool ok = (uint32_t)(mi >> 32) == tag && (uint32_t)(si >> 32) == tag;
#pragma unroll
            for (int i = 0; i < kRows; ++i) ok = ok && (uint32_t)(v[i] >> 32) == tag;
            got = __builtin_amdgcn_ballot_w64(!ok) == 0;
            if ((MODE & 16) && wid != 1 && got) {               // the sentinel says "all there": the columns themselves, one more round trip
                unsigned long long acc = 0;
#pragma unroll
                for (int i = 0; i < kRows; ++i) acc += ld_gran(gran + (size_t)i * kPS + 2 + col);
                asm volatile("" :: "v"(acc));
            }
            te = wall();
            if (!got) { prev2 = prev; prev = te - ts; if (MODE & 4) __builtin_amdgcn_s_sleep(4); }
        }
        if ((MODE & 2) && wid == 1 && lane == 0) *(volatile int *)&flag = g + 1;
        if (lane == 0) {
            const size_t o = ((size_t)g * kRows + p_) * 3 + (wid - 1);
            seen[o] = te;
            polls[o * 3 + 0] = te - ts; polls[o * 3 + 1] = prev; polls[o * 3 + 2] = prev2;
        }
    }
}

__global__ void k_now(unsigned long long *o) { *o = wall_clock64(); }

template <int MODE>
static void run(const char *label)
{
    unsigned long long *gran, *wlast, *seen, *polls;
    (void)hipMalloc(&gran, 8 * 16384); (void)hipMemset(gran, 0, 8 * 16384);
    (void)hipMalloc(&wlast, kGen * kRows * 8); (void)hipMemset(wlast, 0, kGen * kRows * 8);
    (void)hipMalloc(&seen, kGen * kRows * 3 * 8); (void)hipMalloc(&polls, kGen * kRows * 9 * 8);
    (void)hipFuncSetAttribute((const void *)k_pub<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024);
    unsigned long long *d_now; (void)hipMalloc(&d_now, 8);
    // the device clock now: a one-thread kernel
    hipLaunchKernelGGL(k_now, 1, 1, 0, 0, d_now);
    unsigned long long now; (void)hipMemcpy(&now, d_now, 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = now + 20000;            // 200 us from now
    hipLaunchKernelGGL(k_pub<MODE>, 2 * kRows, 320, 81 * 1024, 0, gran, t0, wlast, seen, polls);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> hw(kGen * kRows), hs(kGen * kRows * 3), hp(kGen * kRows * 9);
    (void)hipMemcpy(hw.data(), wlast, hw.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs.data(), seen, hs.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hp.data(), polls, hp.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d[3], last[3], p1[3], p2[3];
    for (int g = 20; g < kGen; ++g) {
        unsigned long long wl = 0;
        for (int r = 0; r < kRows; ++r) wl = std::max(wl, hw[(size_t)g * kRows + r]);
        for (int p = 0; p < kRows; ++p) for (int w = 0; w < 3; ++w) {
            const size_t o = ((size_t)g * kRows + p) * 3 + w;
            d[w].push_back(((double)hs[o] - (double)wl) * 0.01);
            last[w].push_back(hp[o * 3] * 0.01); p1[w].push_back(hp[o * 3 + 1] * 0.01); p2[w].push_back(hp[o * 3 + 2] * 0.01);
        }
    }
    auto med = [](std::vector<double> &v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("%s\n", label);
    const char *names[3] = {"wave 1 (8 columns)", "wave 2 (64 columns)", "wave 3 (36 columns)"};
    for (int w = 0; w < 3; ++w)
        printf("  %-20s rows seen %5.2f us after the last writer's last store was issued (p10 %5.2f, p90 %5.2f) | the poll that saw them %4.2f us, the one before %4.2f, before that %4.2f\n",
               names[w], med(d[w], 0.5), med(d[w], 0.1), med(d[w], 0.9), med(last[w], 0.5), med(p1[w], 0.5), med(p2[w], 0.5));
    (void)hipFree(gran); (void)hipFree(wlast); (void)hipFree(seen); (void)hipFree(polls); (void)hipFree(d_now);
}

int main()
{
    run<0>("the product's pattern: scattered 8-byte stores over ~0.4 us, three polling waves per waiting workgroup");
    run<1>("one burst per row (16-byte stores by one wave at +0.5 us)");
    run<2>("scattered stores; waves 2 and 3 start polling when wave 1 has seen its columns");
    run<4>("scattered stores; ~0.1 us of sleep between polls");
    run<3>("burst + staged polling");
    run<64>("scattered stores + one sentinel per row in its own line; wave 1 polls the sentinels, then everybody reads (tags validate)");
    run<65>("burst + sentinel");
    run<32>("scattered stores; the pollers read ANOTHER array (first figure: give-up time, meaningless; `saw them` = the longest poll within 1.5 us of the publication, `before` = an idle poll)");
    run<33>("burst; the pollers read another array");
    run<16>("scattered stores; waves 2 and 3 poll one sentinel per row, then read their columns");
    run<17>("burst; waves 2 and 3 poll one sentinel per row, then read their columns");
    return 0;
}
