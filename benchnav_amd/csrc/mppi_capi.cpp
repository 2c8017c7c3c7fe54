// mppi_capi.cpp -- host side of the C ABI declared in include/benchnav_mppi.h.
//
// Owns the device buffers of one planner handle and enqueues the two kernels of
// a solve (mppi_kernels.hip) on the handle's HIP stream.  No torch types, no CPU
// fallback: creation fails without a gfx950 device.
#include "../../include/benchnav_mppi.h"
#include "mppi_kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <sys/stat.h>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <dlfcn.h>
#include <hsa/hsa.h>        // types only (as RCCL below): the HSA runtime HIP itself runs on is looked up at run time, and only by a host-paced handle
#include <hsa/hsa_ext_amd.h>
#include <rccl/rccl.h>      // types only: the library itself is opened at run time (rccl_api), and only by a K-sharded solve
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

namespace {

constexpr int kProfGroup = 10;
// Role kernel or one-wave throughput kernel for a many-instance launch?  Both give bit-identical results; which one is faster
// depends on the horizon (the role kernel's skeleton -- prologue, barriers, epilogue -- is a fixed cost per workgroup, the one-wave
// kernel pays per step), on the workgroups per instance (both re-merge nblk rows per workgroup) and on how full the chip gets.
// A two-term estimate of either kernel's time per launch, fitted to tools/auto_sweep.py (K 128..2048, T 10..100, 16..300
// instances, overlapped launches): it picks the faster kernel or one within 10 % of it in 51 of 52 cells (worst: 28 %), where a
// fixed workgroup-count threshold lost 14-60 % in seven.  K=1024, T=50: crossover between 64 and 128 instances as measured.
bool wave_kernel_is_faster(const bn::SolveParams &p, size_t resident_role_wgs, int n_cus)
{
    const double W = (double)p.B * (p.nblk + 1), R = (double)std::max<size_t>(resident_role_wgs, 1), rows = std::max(0, p.nblk - 16);
    const double role = (4.5 + 0.19 * p.T + 0.3 * rows) * (1.0 + 0.175 * std::min(W, R) / std::max(n_cus, 1)) * std::max(1.0, W / R);
    // (round 5: x 0.93 -- the one-wave kernel lost a quarter of its instructions; re-checked against tools/auto_sweep.py, which had
    // flagged K=1024 T=50 at 80 instances and K=512 T=20 at 128 where the role kernel spills into a second residency round)
    // (not for short horizons or more than 16 workgroups per instance, where the sweep has the role kernel ahead at the old crossover)
    const double gain = (rows == 0 && p.T >= 20) ? 0.93 : 1.0;
    const double wave = gain * (3.3 + 0.44 * p.T + 0.37 * rows) * (1.0 + 1.5 * W / (24.0 * std::max(n_cus, 1)));
    return wave < role;
}
thread_local std::string g_last_error;

// Experiment switches (environment variables read by the measurement tools under tools/): compiled in only with -DBN_EXPERIMENTS
// (tools/build_variant*.py pass it); the shipped library has none in its entry points.
#ifdef BN_EXPERIMENTS
inline const char *exp_env(const char *name) { return std::getenv(name); }
#else
inline const char *exp_env(const char *) { return nullptr; }
#endif

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define BN_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError();   /* reported here: do not leave it for the next HIP user of the process (torch) */ \
            return fail(BN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
        }                                                                                    \
    } while (0)

bool is_pow2_float(float v)
{
    int ex;
    return v > 0.0f && std::isfinite(v) && std::frexp(v, &ex) == 0.5f;
}

// One overlapped batch per device at a time.  Workgroups of an overlapped launch hold their slots while they wait for their
// predecessor; two handles doing that at once can starve each other's predecessors of slots until the bounded waits expire
// (8 handles x 8 instances: seen).  A handle overlaps only while no OTHER handle's overlapped batch is still in flight on its
// device (its streams still busy); otherwise this batch runs in one stream, which is always safe.
std::mutex g_overlap_mu;
std::map<int, bn_mppi *> g_overlap_owner;   // device -> the handle whose overlapped batch was enqueued last
// ... and nothing ELSE of this process beside it (round 4).  The rule above kept two overlapped batches apart; a differential sweep
// over pairs of planners (tools/fuzz_features2.py) still saw waits expire: one planner's overlapped batch of latency-kernel
// launches -- one workgroup per CU, sized for a device it has to itself -- with the other planner's ordinary one-stream launches
// in between (12 % of the runs of one pattern, depending on the order the handles' queues were created in).  So a batch
// overlaps only while EVERY other handle on the device is idle (g_handles), and a launch of another handle that arrives while an
// overlapped batch is in flight is ordered behind that batch's end by events (order_behind_foreign_overlap): no host blocking,
// the kernels simply do not share the device with waiting workgroups.
std::map<int, std::vector<bn_mppi *>> g_handles;   // device -> live handles (guarded by g_overlap_mu)
std::atomic<int> g_overlap_owners{0};              // devices with an owner: the launch path looks at it without the lock
// ... and one PROCESS per device for launches big enough to crowd each other out: two processes with overlapped 64-instance batches
// on one GPU both ran into expired waits within a second.  An advisory lock on a per-device file in /tmp, taken (non-blocking) by
// the first process that overlaps big launches there and held until it exits; a process that does not get it runs those batches
// on one stream and asks again at its next batch.
std::map<int, int> g_overlap_lock_fd;       // device -> fd holding the lock (-1: not held)

bool own_device_for_big_overlap(int device)
{
    auto it = g_overlap_lock_fd.find(device);
    if (it != g_overlap_lock_fd.end() && it->second >= 0) return true;
    char bus[64] = "unknown";
    (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, device);
    for (char *c = bus; *c; ++c)
        if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    const std::string path = std::string("/tmp/benchnav_mppi_overlap_") + bus + ".lock";
    // Read-only open (flock works on it): a file another user created 0644 still opens; no symlink followed in a world-writable
    // directory.  The creator widens the mode past its umask so the next user can do the same.
    const int fd = open(path.c_str(), O_CREAT | O_RDONLY | O_NOFOLLOW | O_CLOEXEC, 0666);
    if (fd < 0) return errno == EROFS || errno == ENOENT;   // no lock file POSSIBLE: behave as a lone process; anything else (EACCES, ELOOP ...): do not overlap
    (void)fchmod(fd, 0666);
    if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); return false; }
    g_overlap_lock_fd[device] = fd;
    return true;
}

// Up to this many workgroups per instance every rollout workgroup re-merges the previous solve's partials itself (pipelined mode:
// no merge launch, no ticket round trips); above, the last workgroup of a launch merges (ticket mode).  64 = what the few-rows
// merge takes in one wave; measured at K=4096: 17.1 us per solve pipelined against 21.2 with the ticket merge (tools/k_sweep.py).
constexpr int kPipelinedMaxBlocks = 64;
constexpr int kEagerTailMinBatch = 16;    // overlapped batches of at least this many solves end with their own tail kernel
constexpr int kMaxStreams = 3;            // launches of one overlapped batch in flight at most
constexpr int kSlots = kMaxStreams + 1;   // per-solve buffer slots (see bn_mppi::d_cost)

}  // namespace

struct bn_mppi {
    bn_mppi_config cfg{};
    bn::SolveParams p{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_maps = 1;
    uint64_t solves = 0;
    bool map_set = false, goal_set = false;
    uint64_t map_epoch = 0, map_epoch_at_solve = 0;   // set_map calls so far / at the latest solve: a re-roll needs the map that solve saw
    // Pipelined mode (K <= 2048): one launch per solve.  The launch of solve i merges solve i-1's
    // per-block statistics in every rollout workgroup (warm start) and carries an aux workgroup that
    // writes solve i-1's tail (U*, X*, weights).  `tail_pending` = the latest solve's tail has not been
    // written yet; flush() launches the finish kernel for it.
    bool pipelined = false;
    bool slow_path = false;          // rollout_wave_kernel + stand-alone tail, two launches per solve: horizons beyond the role kernels' LDS, or the reference-order
                                     // arithmetic together with sampled slip (the reference order by itself runs on every kernel family, one launch per solve)
    bool tail_pending = false;
    // device buffers
    float *d_map = nullptr, *d_state = nullptr, *d_goal = nullptr, *d_mean = nullptr, *d_eps = nullptr;
    float *d_X = nullptr, *d_U = nullptr, *d_w = nullptr, *d_cost_out = nullptr;
    // per-solve buffers rotate over kSlots slots (solve i uses slot i % kSlots): with S launches in flight (see bn_mppi_solve_n_async)
    // solve i+S may start while solve i+1 -- its merge, its aux workgroup -- still reads solve i's slot: S + 1 slots are enough
    float *d_cost[kSlots] = {}, *d_part[kSlots] = {}, *d_state_copy[kSlots] = {};
    float *d_ustar = nullptr, *d_xstar = nullptr, *d_stats = nullptr, *d_scratch = nullptr;
    float *d_mean_used = nullptr;    // the mean the latest finished solve sampled around (re-rolls)
    int *d_idx = nullptr;
    const float *last_eps = nullptr; // noise of the latest solve (caller-owned unless BN_NOISE_HOST_KT2) and its layout
    bn::EpsMode last_mode = bn::kEpsPhilox;
    float *d_slip_std = nullptr;     // sampled-slip mode
    float *d_ustar2[kSlots] = {}, *d_stats2[kSlots] = {};   // ticket-merge outputs, one per slot like the other per-solve buffers (overlapped: S + 1 needed)
    bool ticket_overlap = false;     // ticket path (deterministic kernel, K > 4096): consecutive launches of a batch overlap too (round 3)
    std::vector<float> dwa_stage;    // host staging of bn_mppi_dwa_solve's upload
    int *d_ticket = nullptr;
    float *d_gpart = nullptr;
    int n_cus = 256;
    size_t resident_wgs = 1024;      // role-kernel workgroups the device holds at once (LDS- and wave-limited) x CUs
    // overlapped launches: consecutive solves of one bn_mppi_solve_n_async call go round the handle's stream and n_streams - 1
    // extra ones; device counters carry the dependency (SolveParams.flag_part / flag_tail)
    int n_streams = 1;
    hipStream_t xstream[kMaxStreams - 1] = {};
    std::vector<hipStream_t> parked;            // streams that turned out to share the handle's hardware queue (see bn_mppi_create)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams - 1] = {};
    hipEvent_t ev_guard[kMaxStreams] = {};      // recorded on this handle's streams BY another handle that must run behind this one's overlapped batch
    unsigned long long *d_flags = nullptr;      // [kSlots][B] flag_part per slot and instance, [B] flag_tail per instance, then int err
    bool role_overlap = false;                  // the role kernel's launches of a batch may overlap too (see bn_mppi_solve_n_async)
    unsigned long long *d_gran[kSlots] = {};   // granule copies of the partial rows per slot (K <= 1024, 2T <= 320)
    // Second trajectory / control buffers for overlapped batches.  Two launches in flight must not write the same addresses: whose
    // dirty L2 lines reach memory last is decided by which KERNEL ends last, and a straggling aux workgroup can make the earlier
    // solve's kernel the later one to end (seen once per cold start: X / U of the last solve partly overwritten by its
    // predecessor's).  Solves of a batch go round n_streams buffers such that the LAST one writes the exposed buffer.
    float *d_Xalt[kMaxStreams - 1] = {}, *d_Ualt[kMaxStreams - 1] = {};
    unsigned long long pub[kSlots] = {};     // host mirror: what flag_part[slot][b] reaches once every launch issued so far has published
    unsigned long long tails = 0;               // host mirror of flag_tail[b]
    bool prev_published = false;                // the latest solve counted itself into flag_part (latency kernel): its successor may overlap
    bool overlap_used = false;                  // a wait could have expired since the last check of the error word
    bool overlap_off = false;                   // a wait DID expire once: this handle keeps to one stream from then on
    // The error word of the bounded device-side waits: pinned host memory the kernels write with a system-scope store, so every
    // entry point can look at it for free.  Journal: the batches enqueued since the word was last found clean at a
    // synchronisation point, with the mean the first of them started from (kept by that launch itself, SolveParams::mean_snap)
    // and the solve counter (= Philox position) at that point -- what recover_overlap() needs to run them again on one stream.
    int *h_err = nullptr, *d_err = nullptr;     // host address / device address of the same word
    unsigned long long *h_mail = nullptr, *d_mail = nullptr;   // (B, 2) granules {U*[0][d], solve index + 1}: see bn_mppi_first_action
    float *d_mean_snap = nullptr;
    float *d_state_snaps = nullptr;             // [snap_cap][B][3]: the states every journalled batch started from, kept by its first launch
    size_t snap_cap = 0;                        // (SolveParams::state_snap): a re-run reads these, not the caller's buffer
    size_t snap_slot = 0;                       // slot of the batch being enqueued
    bool arm_state_snap = false;                // the next launch fills it
    struct BatchRec { int32_t n; const float *states; const float *eps; bn_noise_kind noise; int32_t eps_ring; int64_t eps_stride; bool episode; const float *z; };
    std::vector<BatchRec> journal;
    bool journal_lost = false;                  // something not replayable happened since the last clean check (or too many batches)
    bool arm_snap = false;                      // the next launch keeps its mean in d_mean_snap
    bool replaying = false, last_batch_overlapped = false;
    uint64_t journal_solves0 = 0, recoveries = 0;
    // Self-protecting overlap (round 6).  Overlapped launches assume the device to themselves: with a co-tenant process on the GPU the
    // workgroups that wait for their predecessor hold slots the predecessor's stragglers need, and the chain runs at HALF the
    // one-stream rate (46 k against 80 k solves/s).  The handle watches its own cadence -- the first-action mailbox
    // counts finished solves, the host clock does the rest: no event, no kernel change -- over windows of 64 launches that found the
    // device backlogged at both ends, looks at the other mode once, and runs whichever is faster; it looks again when its cadence
    // degrades by half, and every 256 windows while it runs on one stream (has the co-tenant left?).
    struct OvTune { double best[2] = {0, 0}, last[2] = {0, 0}, ema[2] = {0, 0}; int chosen = 0, explore = 0, slow_run = 0; uint64_t windows[2] = {0, 0}, since_explore = 0, switches = 0; } tune;
    struct OvSample { bool valid = false; std::chrono::steady_clock::time_point t{}; uint32_t prog = 0, enq = 0; } ov_sample;
    int run_mode = 0;                // the mode of the batch being enqueued (0 overlapped, 1 one stream)
    // One launch per SYNCHRONOUS solve (round 6; bn_mppi_forward_async, bn_mppi_forward_state_async, bn_mppi_solve on the latency kernel):
    // the solve's own tail rides in its launch as a second aux workgroup (SolveParams::self_tail) -- no stand-alone finish kernel.
    float *self_out_copy = nullptr;  // the caller's U* | X* block of the forward() being enqueued (consumed by solve_impl)
    const float *inline_state = nullptr;   // host pointer: the state of the forward() being enqueued travels in the kernel arguments
    bool self_tail_launched = false; // the latest solve_impl call took the one-launch path (its tail is NOT pending)
    bool self_used = false;          // a self tail's bounded wait could have expired since the error word was last looked at
    bool self_off = false;           // ... and one did: two launches per synchronous solve from then on
    // Host-paced loop (BN_FLAG_HOST_PACED, round 6): bn_mppi_forward_state_async enqueues the NEXT solve's launch one control step ahead,
    // on one of two private streams; that launch waits on the device for the state the next call posts (rollout_lat.inc, HOSTP).
    bool hp_enabled = false;
    hipStream_t hp_stream[2] = {};   // [0] = xstream[0], [1] = xstream[1] (created for this mode); the launches alternate
    hipEvent_t hp_ev[2] = {};        // recorded behind every prelaunch; the handle's stream waits for it when the request is posted
    bool hp_unordered = false;       // BN_FLAG_UNORDERED_OUTPUTS: the handle's stream has not been ordered behind the latest posted solve yet ...
    int hp_unordered_q = 0;          // ... whose launch is the latest one on this private stream (see hp_order_now)
    unsigned long long *h_req = nullptr, *d_req = nullptr;         // pinned: kSlots x 8 request granules, then 16 status words (acknowledgements, "gave up" per slot)
    unsigned long long *req_bar = nullptr;                         // the request granules in DEVICE memory the host writes through the BAR (bar_alloc), or null
    unsigned long long *d_req_dev = nullptr;                       // device: kSlots x 8, republished by the launches' tail workgroups
    uint32_t hp_seq = 0;             // request tags, unique per prelaunch
    bool hp_armed = false;           // a prelaunched solve waits for its state
    uint32_t hp_tag = 0;             // ... its tag, request slot, stream index, trajectory buffer
    int hp_slot = 0, hp_q = 0, hp_xidx = 0;
    uint32_t hp_posted_tag = 0;      // the latest request posted (bn_mppi_first_action watches the give-up word for it)
    float hp_posted_state[3] = {};
    float *hp_posted_out = nullptr;
    bool hp_skip_check = false;      // test hook: post without looking whether the launch has given up (exercises the repair in bn_mppi_first_action)
    float hp_prev_state[3] = {};     // the state of the latest forward (the next launch's speculative window is staged around it)
    bool hp_gran_valid = false;      // the latest solve published its partial rows as granules and nothing has touched the mean since
    int hp_extra = 2;                // cells the speculative window is wider on each side
    int x_idx = 0;                   // which trajectory / control buffer holds the latest solve (bn_mppi_states_buffer_index)
    int hp_next_q = 0;
    std::chrono::steady_clock::time_point hp_armed_at{};      // when the waiting launch was enqueued (a request is posted only well within its patience)
    int hp_polls = 25000;            // how long a prelaunched solve waits for its state: looks of ~2 us each (~50 ms); bn_mppi_debug_host_paced
    bool lat_kernel = false;         // plain pipelined solves of a launch that leaves every workgroup a CU: rollout_lat_kernel
    bool wave_kernel = false;        // plain pipelined solves use rollout_wave_kernel (episodes keep the role kernel)
    bool shard_pending = false;      // K-sharded solve: rollouts launched, tail waits for the partials of the other shards
    void *shard_comm = nullptr;      // ncclComm_t of bn_mppi_shard_comm_init: the exchange is enqueued by the library on the handle's stream
    int shard_world = 0, shard_rank = 0;
    bool shard_prepared = false;     // bn_mppi_shard_comm_prepare has run: buffers, events, side stream exist; the communicator may not yet
    bool shard_inplace = false;               // set around the rollouts of bn_mppi_shard_solve_async
    float *d_gathered = nullptr;     // (shard_world x nblk, 2 + 2T): every shard's partial rows, rank order
    float *d_shard_merged = nullptr; // kSlots x [U* (2T) then (max z, sum e)]: what the merge kernel hands to the tail on the side stream
    hipStream_t shard_side = nullptr;         // the tail of a library-enqueued sharded solve runs here, beside the next solve's rollouts
    bool shard_side_own = false;
    hipEvent_t ev_shard_merge = nullptr, ev_shard_tail[kSlots] = {};      // tail events by solve slot (see bn_mppi_shard_solve_async)
    int shard_tail_slot = 0;                  // slot of the latest tail
    bool shard_tail_inflight = false;         // the handle's stream has not been ordered behind the latest side-stream tail yet
    bool ticket_mode = false;        // one launch per solve: ticket merge by the last workgroup + the previous tail as aux
                                     // workgroup (sampled-slip kernel; deterministic kernel at K > 2048)
    bool slip_std_set = false;
    // device-side closed loop (bn_mppi_env_attach / bn_mppi_episode_async)
    float *d_lat_mean = nullptr, *d_lat_std = nullptr, *d_ep_states = nullptr, *d_ep_reward = nullptr, *d_env_state = nullptr;
    float *d_ep_action = nullptr;
    int *d_ep_done = nullptr;
    bool env_attached = false;
    int ep_steps = 0;            // capacity of the episode log
    int ep_len = 0;              // steps enqueued in the current episode
    const float *ep_z = nullptr; // (n_steps, B) device slip draws of the current episode, or nullptr
    bool in_episode = false;
    size_t scratch_bytes = 0, eps_bytes = 0, idx_count = 0;
    float *h_pinned = nullptr;   // pinned staging for (B,3) states
    // profiling
    std::vector<hipEvent_t> ev;  // two-launch mode: 3 events per solve (start, after rollout, after finish);
    size_t ev_used = 0;          // pipelined mode: one event at the start of every group of kProfGroup launches
    int prof_in_group = 0;
};

namespace {

// Every entry point works on the handle's device and leaves the calling thread's current device as it found it
// (one process may drive several GPUs; torch tracks its own notion of the current device).
struct DeviceGuard {
    int prev = -1;
    bool changed = false, ok = true;
    explicit DeviceGuard(int want)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != want) {
            ok = hipSetDevice(want) == hipSuccess;
            changed = ok;
        }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define BN_BIND_Q(h) DeviceGuard bn_guard_((h)->cfg.device_id); if (!bn_guard_.ok) return fail(BN_ERR_HIP, "hipSetDevice failed")

// BN_FLAG_UNORDERED_OUTPUTS: the host-paced forward left the handle's stream unordered behind the latest posted solve (a stream-wait on a
// pending event is 3-5 us of host time per control step, for consumers that mostly are not there).  Every entry point that enqueues on the
// handle's stream or hands out results makes up for it here, once -- everything except the loop's own two calls.
static inline void hp_order_now(bn_mppi *h)
{
    if (!h->hp_unordered) return;
    h->hp_unordered = false;
    if (hipStreamWaitEvent(h->stream, h->hp_ev[h->hp_unordered_q], 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->hp_stream[h->hp_unordered_q]);
    }
}
#define BN_BIND(h) BN_BIND_Q(h); hp_order_now(h)

size_t buffer_bytes(const bn_mppi *h, bn_buffer_id id)
{
    const size_t B = h->p.B, K = h->p.K, T = h->p.T, G = h->p.G;
    switch (id) {
    case BN_BUF_STATES: return B * (T + 1) * 3 * (size_t)h->p.Kp * 4;
    case BN_BUF_STATES_ALT: return h->d_Xalt[0] ? B * (T + 1) * 3 * (size_t)h->p.Kp * 4 : 0;
    case BN_BUF_CONTROLS_ALT: return h->d_Ualt[0] ? B * T * 2 * (size_t)h->p.Kp * 4 : 0;
    case BN_BUF_WEIGHTS: return B * K * 4;
    case BN_BUF_COSTS: return B * K * 4;
    case BN_BUF_CONTROLS: return h->d_U ? B * T * 2 * (size_t)h->p.Kp * 4 : 0;
    case BN_BUF_USTAR: return B * T * 2 * 4;
    case BN_BUF_XSTAR: return B * (T + 1) * 3 * 4;
    case BN_BUF_MEAN: return B * T * 2 * 4;
    case BN_BUF_MAP: return (size_t)h->n_maps * G * G * 4;
    case BN_BUF_GOAL: return B * 2 * 4;
    case BN_BUF_USTAR_XSTAR: return (B * T * 2 + B * (T + 1) * 3) * 4;
    default: return 0;
    }
}

int ensure_scratch(bn_mppi *h, size_t bytes)
{
    if (bytes <= h->scratch_bytes) return BN_OK;
    if (h->d_scratch) BN_HIP(hipFree(h->d_scratch));
    h->d_scratch = nullptr;
    h->scratch_bytes = 0;
    BN_HIP(hipMalloc(&h->d_scratch, bytes));
    h->scratch_bytes = bytes;
    return BN_OK;
}

int guard_foreign_overlap(bn_mppi *h);      // (defined with order_behind_foreign_overlap)

// ---- host-paced loop: request words the HOST writes straight into device memory ---------------------------------------------------
// The waiting launch polls its request.  Pinned host memory makes every look a PCIe read round trip (~3 us on this platform: the
// state was seen ~4 us after the host's store, a quarter of the control step).  Device memory the CPU can write -- a fine-grained VRAM
// allocation of the HSA runtime HIP itself runs on, opened to the CPU agent: stores go out through the PCIe BAR as posted writes --
// turns that around: the host's store travels once, the GPU polls its own memory.  No such allocation (no large BAR, an HSA runtime
// without the entry points, an agent that cannot be matched to the HIP device): the words stay in pinned host memory, as before.
struct HsaApi {
    void *lib = nullptr;
    decltype(&hsa_iterate_agents) iterate_agents = nullptr;
    decltype(&hsa_agent_get_info) agent_get_info = nullptr;
    decltype(&hsa_amd_agent_iterate_memory_pools) iterate_pools = nullptr;
    decltype(&hsa_amd_memory_pool_get_info) pool_get_info = nullptr;
    decltype(&hsa_amd_memory_pool_allocate) pool_allocate = nullptr;
    decltype(&hsa_amd_memory_pool_free) pool_free = nullptr;
    decltype(&hsa_amd_agents_allow_access) allow_access = nullptr;
};
const HsaApi *hsa_api()
{
    static const HsaApi api = [] {
        HsaApi a;
        for (const char *name : {"libhsa-runtime64.so.1", "libhsa-runtime64.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);      // the copy HIP has loaded, or none
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.iterate_agents = reinterpret_cast<decltype(a.iterate_agents)>(dlsym(a.lib, "hsa_iterate_agents"));
        a.agent_get_info = reinterpret_cast<decltype(a.agent_get_info)>(dlsym(a.lib, "hsa_agent_get_info"));
        a.iterate_pools = reinterpret_cast<decltype(a.iterate_pools)>(dlsym(a.lib, "hsa_amd_agent_iterate_memory_pools"));
        a.pool_get_info = reinterpret_cast<decltype(a.pool_get_info)>(dlsym(a.lib, "hsa_amd_memory_pool_get_info"));
        a.pool_allocate = reinterpret_cast<decltype(a.pool_allocate)>(dlsym(a.lib, "hsa_amd_memory_pool_allocate"));
        a.pool_free = reinterpret_cast<decltype(a.pool_free)>(dlsym(a.lib, "hsa_amd_memory_pool_free"));
        a.allow_access = reinterpret_cast<decltype(a.allow_access)>(dlsym(a.lib, "hsa_amd_agents_allow_access"));
        if (!a.iterate_agents || !a.agent_get_info || !a.iterate_pools || !a.pool_get_info || !a.pool_allocate || !a.pool_free || !a.allow_access) a.lib = nullptr;
        return a;
    }();
    return api.lib ? &api : nullptr;
}

struct BarFind { const HsaApi *api; uint32_t bdf, domain; hsa_agent_t gpu{}, cpu{}; bool have_gpu = false, have_cpu = false; hsa_amd_memory_pool_t pool{}; int pool_rank = 0; };

hsa_status_t bar_agent_cb(hsa_agent_t agent, void *data)
{
    BarFind *f = static_cast<BarFind *>(data);
    hsa_device_type_t type;
    if (f->api->agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (type == HSA_DEVICE_TYPE_CPU) { if (!f->have_cpu) { f->cpu = agent; f->have_cpu = true; } return HSA_STATUS_SUCCESS; }
    if (type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    if (f->api->agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)f->api->agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == f->bdf && dom == f->domain) { f->gpu = agent; f->have_gpu = true; }
    return HSA_STATUS_SUCCESS;
}

hsa_status_t bar_pool_cb(hsa_amd_memory_pool_t pool, void *data)
{
    BarFind *f = static_cast<BarFind *>(data);
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    if (f->api->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    if (f->api->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc) != HSA_STATUS_SUCCESS || !alloc) return HSA_STATUS_SUCCESS;
    (void)f->api->pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    // fine-grained first (coherent with the host by construction), then the extended-scope kind; never the coarse-grained pool
    const int rank = (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) ? 2 : (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED) ? 1 : 0;
    if (rank > f->pool_rank) { f->pool = pool; f->pool_rank = rank; }
    return HSA_STATUS_SUCCESS;
}

// `bytes` of device memory the CPU may write; nullptr when the platform does not offer it.
void *bar_alloc(int device_id, size_t bytes)
{
    const HsaApi *a = hsa_api();
    if (!a) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    BarFind f;
    f.api = a;
    f.bdf = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
    f.domain = (uint32_t)prop.pciDomainID;
    if (a->iterate_agents(bar_agent_cb, &f) != HSA_STATUS_SUCCESS || !f.have_gpu || !f.have_cpu) return nullptr;
    if (a->iterate_pools(f.gpu, bar_pool_cb, &f) != HSA_STATUS_SUCCESS || f.pool_rank == 0) return nullptr;
    void *ptr = nullptr;
    if (a->pool_allocate(f.pool, (bytes + 4095) & ~(size_t)4095, 0, &ptr) != HSA_STATUS_SUCCESS || !ptr) return nullptr;
    if (a->allow_access(1, &f.cpu, nullptr, ptr) != HSA_STATUS_SUCCESS) { (void)a->pool_free(ptr); return nullptr; }
    return ptr;
}

void bar_free(void *ptr)
{
    if (!ptr) return;
    if (const HsaApi *a = hsa_api()) (void)a->pool_free(ptr);
}

// ---- host-paced loop: the request words --------------------------------------------------------------------------------------
// A request is six {value, tag} granules in pinned memory (x, y, theta, the caller's output block lo / hi, the command); the
// launch's tail workgroup takes them when all six carry the launch's tag.  Plain 8-byte stores: x86 keeps them in order, and each
// granule vouches for itself anyway.
void hp_write_request(bn_mppi *h, int slot, uint32_t tag, const float st[3], const float *out, uint32_t cmd)
{
    unsigned long long *r = (h->req_bar ? h->req_bar : h->h_req) + (size_t)slot * 8;
    uint32_t w[6] = {0, 0, 0, 0, 0, cmd};
    if (st) std::memcpy(w, st, 12);
    const unsigned long long o = (unsigned long long)(uintptr_t)out;
    w[3] = (uint32_t)o; w[4] = (uint32_t)(o >> 32);
    for (int i = 0; i < 6; ++i) __atomic_store_n(r + i, ((unsigned long long)tag << 32) | w[i], __ATOMIC_RELEASE);
    if (h->req_bar) __builtin_ia32_sfence();           // device memory through the BAR is write-combining: out with it now
}

// The tail workgroup of the launch with this tag gave up waiting for the host (~50 ms) and said so: the launch has ended, nothing happened.
bool hp_gave_up(const bn_mppi *h, uint32_t tag)
{
    return tag != 0 && h->h_req && __atomic_load_n(h->h_req + (size_t)kSlots * 8 + 8 + (tag % kSlots), __ATOMIC_ACQUIRE) == (((unsigned long long)tag << 32) | 2ull);
}

// A launch that still waits for its state is told to leave -- it has touched nothing but its own LDS -- and the host's bookkeeping
// goes back to where it was before the prelaunch.  No synchronisation: the launch ends by itself within a poll.
void hp_cancel(bn_mppi *h)
{
    if (!h->hp_armed) return;
    hp_write_request(h, h->hp_slot, h->hp_tag, nullptr, nullptr, 2u);
    h->hp_armed = false;
    h->solves -= 1;
    const int cur3 = (int)(h->solves % kSlots);
    h->pub[cur3] -= (unsigned long long)h->p.nblk;
    h->tails -= 1;
    h->hp_next_q = h->hp_q;                            // (the stream the cancelled launch leaves within a poll)
}

// Write the tail (U*, next mean, X*, weights, cost copy) of the latest solve if it is still pending.
int flush_tail(bn_mppi *h, float *out_copy = nullptr)
{
    if (h->shard_tail_inflight) {                      // K-sharded solve, library-enqueued: results are ordered on the handle's stream from here on
        BN_HIP(hipStreamWaitEvent(h->stream, h->ev_shard_tail[h->shard_tail_slot], 0));
        h->shard_tail_inflight = false;
    }
    if (!h->tail_pending) return BN_OK;
    bn::SolveParams p = h->p;
    p.out_copy = out_copy;
    const int cur3 = (int)((h->solves - 1) % kSlots);
    p.part = h->d_part[cur3]; p.cost = h->d_cost[cur3]; p.state = h->d_state_copy[cur3];
    if (h->ticket_mode) {
        p.tail_merged = 1;
        p.ustar_prev = h->d_ustar2[cur3]; p.stats_prev = h->d_stats2[cur3];
    }
    p.tail_solve = h->solves - 1;
    if (h->in_episode) {
        p.env_on = 1;
        p.ep_index = h->ep_len - 1;
        p.env_z = h->ep_z ? h->ep_z + (size_t)(h->ep_len - 1) * p.B : nullptr;
    }
    p.flag_tail = h->d_flags + kSlots * (size_t)p.B * bn::kFlagStride;        // every tail counts itself in (stream-ordered here: nothing to wait for)
    h->tails += 1;
    if (h->overlap_used) p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * (size_t)p.B + 2) * bn::kFlagStride);   // behind overlapped launches: see raise_wait_expired
    if (int rc = guard_foreign_overlap(h)) return rc;
    BN_HIP(bn::launch_finish(p, h->stream));
    h->tail_pending = false;
    return BN_OK;
}

int check_instance(const bn_mppi *h, int32_t instance, bool allow_all)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (instance == -1 && allow_all) return BN_OK;
    if (instance < 0 || instance >= h->p.B)
        return fail(BN_ERR_INVALID, "instance %d out of range [0,%d)", instance, h->p.B);
    return BN_OK;
}

constexpr size_t kJournalCap = 4096;

// Called before the first launch of a batch that journal_push may record: that launch copies the states it reads into the slot
// the record will point at.  (Costs nothing on the host; one predicated 12-byte store in the kernel.)
void journal_arm_states(bn_mppi *h)
{
    if (h->replaying) return;
    h->snap_slot = h->journal.size();
    h->arm_state_snap = h->snap_slot < h->snap_cap;
}

void journal_push(bn_mppi *h, bn_mppi::BatchRec r, bool replayable)
{
    if (h->replaying) return;
    if (!h->last_batch_overlapped && h->journal.empty()) return;      // nothing in flight that a wait could have spoilt
    if (!replayable || h->journal.size() >= std::min(kJournalCap, h->snap_cap) || h->snap_slot != h->journal.size()) { h->journal_lost = true; h->journal.clear(); return; }
    r.states = h->d_state_snaps + h->snap_slot * (size_t)h->p.B * 3;  // what the batch's first launch kept (journal_arm_states)
    if (!h->journal_lost) h->journal.push_back(r);
}

// A bounded device-side wait of an overlapped launch expired: that launch computed on incomplete partials, and every solve
// warm-started from it since is invalid as well.  Bring the handle to rest, forget the counters, and run the journalled batches
// again on ONE stream from the mean the first of them started from: same inputs (the states each batch's first launch kept in
// d_state_snaps; the callers' noise buffers, which must stay valid until the synchronisation point that follows a batch, as for
// any asynchronous call), same Philox positions, hence
// the results the overlapped launches would have produced.  The handle keeps to one stream from then on: whatever kept a
// predecessor from becoming resident (another process on the GPU, most likely) may still be there.
// BN_OK with a warning in bn_last_error() when the re-run succeeded (bn_mppi_recovery_count counts them: consumers enqueued
// in stream order BEFORE this synchronisation point have read invalid buffers); BN_ERR_HIP when there was nothing to re-run from.
int recover_overlap(bn_mppi *h)
{
    for (int q = 0; q < kMaxStreams - 1; ++q)
        if (h->xstream[q]) BN_HIP(hipStreamSynchronize(h->xstream[q]));
    BN_HIP(hipStreamSynchronize(h->stream));
    BN_HIP(hipMemset(h->d_flags, 0, ((kSlots + 1) * (size_t)h->p.B + 3) * bn::kFlagStride * sizeof(unsigned long long)));      // counters and the device error word
    // ... and the ticket counters of the ticket merge (K > 4096, sampled slip): they are per instance, not per launch -- a launch that
    // gave up waiting and the starved predecessor it gave up on draw from them at the same time, and the merger's reset to zero races
    // with the other launch's increments.  Left non-zero they make the re-run's first launch merge before its rows are complete: a
    // repair with wrong results (round 4, tests/differential.py `ops` seed 2681: half of the natural expiries of that configuration).
    if (h->d_ticket) BN_HIP(hipMemset(h->d_ticket, 0, (size_t)h->p.B * 65 * 4));
    for (int q = 0; q < kSlots; ++q) h->pub[q] = 0;
    h->tails = 0;
    h->prev_published = false;
    h->tail_pending = false;
#ifdef BN_EXPERIMENTS
    fprintf(stderr, "[bn] expired wait: counter %d (B=%d: [slot][b] below %d, tails above) need %d seen %d | solve %d block (%d,%d) have_prev/overlap/cur_slot %d wait_part*1000+wait_tail %d | host: solves %llu tails %llu pub %llu %llu %llu %llu\n",
            h->h_err[8], h->p.B, kSlots * h->p.B, h->h_err[9], h->h_err[10], h->h_err[11], h->h_err[12], h->h_err[13], h->h_err[14], h->h_err[15],
            (unsigned long long)h->solves, (unsigned long long)h->tails, (unsigned long long)h->pub[0], (unsigned long long)h->pub[1], (unsigned long long)h->pub[2], (unsigned long long)h->pub[kSlots - 1]);
#endif
    h->overlap_off = true;
    h->overlap_used = false;
    h->arm_snap = false;
    h->arm_state_snap = false;
    h->in_episode = false;
    *h->h_err = 0;
    std::vector<bn_mppi::BatchRec> recs;
    recs.swap(h->journal);
    const bool lost = h->journal_lost || recs.empty();
    h->journal_lost = false;
    if (lost)
        return fail(BN_ERR_HIP, "an overlapped launch gave up waiting for its predecessor's partials and the batches since the last "
                                "synchronisation point cannot be re-run: their results are invalid; this handle runs its launches on one "
                                "stream from now on");
    BN_HIP(hipMemcpy(h->d_mean, h->d_mean_snap, (size_t)h->p.B * h->p.T * 2 * 4, hipMemcpyDeviceToDevice));
    h->solves = h->journal_solves0;
    h->replaying = true;
    int rc = BN_OK;
    for (const auto &r : recs) {
        rc = r.episode ? bn_mppi_episode_async(h, r.n, r.states, BN_MEM_DEVICE, r.eps, r.noise, r.eps_ring, r.eps_stride, r.z)
                       : bn_mppi_solve_n_async(h, r.n, r.states, BN_MEM_DEVICE, r.eps, r.noise, r.eps_ring, r.eps_stride);
        if (rc != BN_OK) break;
    }
    if (rc == BN_OK) rc = flush_tail(h);
    h->replaying = false;
    if (rc != BN_OK) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    h->recoveries += 1;
    (void)fail(BN_OK, "warning: an overlapped launch gave up waiting for its predecessor's partials; %zu batch(es) were re-run on one "
                      "stream (results are valid now; this handle no longer overlaps its launches)", recs.size());
    return BN_OK;
}

// The handle's stream has just been synchronised (`synced`), or the caller only wants to know what has ALREADY gone wrong:
// look at the error word, repair if it is set.
int settle_overlap(bn_mppi *h, bool synced)
{
    if (!h->overlap_used || h->replaying) return BN_OK;
    if (__atomic_load_n(h->h_err, __ATOMIC_ACQUIRE) == 0) {
        if (synced) {
            bool idle = true;                                  // the extra streams are joined into the handle's stream; make sure
            for (int q = 0; idle && q + 1 < h->n_streams; ++q) idle = hipStreamQuery(h->xstream[q]) == hipSuccess;
            (void)hipGetLastError();
            if (idle) { h->overlap_used = false; h->journal.clear(); h->journal_lost = false; }
        }
        return BN_OK;
    }
    return recover_overlap(h);
}

int sync_checked(bn_mppi *h)
{
    BN_HIP(hipStreamSynchronize(h->stream));
    // the extra streams of an overlapped batch are not joined into the handle's stream (see bn_mppi_solve_n_async): their last
    // kernels have done all their work by now -- the handle's stream consumed it -- and end within microseconds
    if (h->overlap_used)
        for (int q = 0; q + 1 < h->n_streams; ++q)
            if (h->xstream[q]) BN_HIP(hipStreamSynchronize(h->xstream[q]));
    return settle_overlap(h, true);
}

// The tail of a one-launch synchronous solve waits (bounded) for the rollout workgroups of its OWN launch.  They are dispatched before
// it and wait for nobody, so the wait ends -- as surely as a barrier does; should it ever expire (the error word, looked at here for
// free) the latest solve's outputs are invalid and there is nothing to re-run from: report it, and keep to two launches from then on.
int self_check(bn_mppi *h)
{
    if (!h->self_used || h->overlap_used || h->replaying) return BN_OK;      // (behind overlapped batches: settle_overlap looks at the same word)
    if (__atomic_load_n(h->h_err, __ATOMIC_ACQUIRE) == 0) return BN_OK;
    BN_HIP(hipStreamSynchronize(h->stream));
    BN_HIP(hipMemset(h->d_flags, 0, ((kSlots + 1) * (size_t)h->p.B + 3) * bn::kFlagStride * sizeof(unsigned long long)));
    for (int q = 0; q < kSlots; ++q) h->pub[q] = 0;
    h->tails = 0;
    *h->h_err = 0;
    h->self_used = false;
    h->self_off = true;
    return fail(BN_ERR_HIP, "the tail of a one-launch solve gave up waiting for the rollout workgroups of its own launch: the outputs of solve %llu "
                            "are invalid (solve again); this handle goes back to two launches per synchronous solve",
                (unsigned long long)h->solves - 1);
}

// Entry points that hand results to the host, change the planner's inputs, or enqueue work the journal does not describe: whatever
// an expired wait could have spoilt is repaired first.  Free unless overlapped launches are outstanding (then: one synchronisation).
int settle_point(bn_mppi *h)
{
    hp_cancel(h);                                      // (a launch waiting for the host's next state: this call is not that state)
    h->hp_gran_valid = false;
    if (int rc = self_check(h)) return rc;
    if (!h->overlap_used || h->replaying) return BN_OK;
    if (int rc = flush_tail(h)) return rc;
    return sync_checked(h);
}

void tune_record(bn_mppi *h, int mode, double us)
{
    bn_mppi::OvTune &t = h->tune;
    t.last[mode] = us;
    t.ema[mode] = t.ema[mode] == 0 ? us : t.ema[mode] + (us - t.ema[mode]) / 8;
    // the mode's best cadence: the minimum over windows that are not implausibly fast (one short window -- a host hiccup between two
    // looks at the mailbox -- must not become the yardstick everything after it fails)
    if (us >= 0.7 * t.ema[mode] && (t.best[mode] == 0 || us < t.best[mode])) t.best[mode] = us;
    t.windows[mode] += 1;
    if (mode == t.chosen) {
        t.since_explore += 1;
        t.slow_run = (t.best[mode] > 0 && us > 1.5 * t.best[mode]) ? t.slow_run + 1 : 0;
        if (t.chosen == 0) {
            if (t.best[1] == 0) { if (t.windows[0] >= 64) { t.explore = 1; t.since_explore = 0; } }      // never seen the other mode: look once (4096 launches in)
            else if (t.slow_run >= 4) {                    // four windows in a row at 1.5 x this mode's best: somebody else is on the device
                t.slow_run = 0;
                if (us > 1.15 * t.best[1]) { t.chosen = 1; t.switches += 1; t.since_explore = 0; }   // ... and one stream was faster than this when last seen
                else if (t.since_explore >= 64) { t.explore = 1; t.since_explore = 0; }
            }
        } else if (t.since_explore >= 256) { t.explore = 1; t.since_explore = 0; }                 // has the co-tenant left?
    } else {                                           // a look at the other mode
        if (t.last[t.chosen] > 0 && us < 0.9 * t.last[t.chosen]) { t.chosen = mode; t.switches += 1; t.since_explore = 0; t.slow_run = 0; }
        t.explore = 0;
    }
}

// Every 64 launches of a batch: how many solves has the device finished (the tails post the first action with the solve's index), and when.
// Two samples that both found the device behind the host give a cadence; anything else says nothing.
void tune_sample(bn_mppi *h)
{
    if (!h->h_mail || h->in_episode || h->replaying) return;
    bn_mppi::OvSample now;
    now.valid = true;
    now.t = std::chrono::steady_clock::now();
    now.prog = (uint32_t)(__atomic_load_n(h->h_mail, __ATOMIC_ACQUIRE) >> 32);
    now.enq = (uint32_t)h->solves;
    const bn_mppi::OvSample &pr = h->ov_sample;
    if (pr.valid && (int32_t)(pr.enq - now.prog) > 0 && (int32_t)(now.prog - pr.prog) >= 16 && (int32_t)(pr.enq - pr.prog) >= 2) {
        const double us = std::chrono::duration<double, std::micro>(now.t - pr.t).count() / (double)(now.prog - pr.prog);
        tune_record(h, h->run_mode, us);
    }
    h->ov_sample = now;
}

}  // namespace

extern "C" {

void bn_mppi_config_init(bn_mppi_config *c)
{
    std::memset(c, 0, sizeof *c);
    c->struct_size = sizeof *c;
    c->horizon = 50;
    c->num_samples = 1024;
    c->num_instances = 1;
    c->grid_size = 64;
    c->resolution = 0.5f;
    c->x_limits[0] = c->y_limits[0] = 0.0f;
    c->x_limits[1] = c->y_limits[1] = 32.0f;
    c->sigma[0] = c->sigma[1] = 0.5f;
    c->inv_var[0] = c->inv_var[1] = 4.0f;
    c->lambda_ = 0.5f;
    c->u_min[0] = 0.0f; c->u_min[1] = -1.0f;
    c->u_max[0] = 1.0f; c->u_max[1] = 1.0f;
    c->dt = 0.1f;
    c->stuck_threshold = 0.3f;
    c->seed = 42;
}

int bn_mppi_create(const bn_mppi_config *cfg, bn_mppi_t **out)
{
    if (!cfg || !out) return fail(BN_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(bn_mppi_config))
        return fail(BN_ERR_INVALID, "bn_mppi_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(bn_mppi_config));
    if (cfg->horizon < 1 || cfg->num_samples < 1 || cfg->num_instances < 1 || cfg->grid_size < 1)
        return fail(BN_ERR_INVALID, "horizon, num_samples, num_instances and grid_size must be >= 1");
    if (cfg->num_instances > 32768) return fail(BN_ERR_INVALID, "num_instances must be <= 32768");
    if ((cfg->flags & BN_FLAG_LEAN) && (cfg->flags & BN_FLAG_SAMPLED_SLIP))
        return fail(BN_ERR_INVALID, "BN_FLAG_LEAN is not available in sampled-slip mode");
    if (!(cfg->resolution > 0.0f) || !(cfg->lambda_ > 0.0f) || !(cfg->dt > 0.0f))
        return fail(BN_ERR_INVALID, "resolution, lambda_ and dt must be positive");
    for (int d = 0; d < 2; ++d)
        if (!(cfg->u_min[d] <= cfg->u_max[d]) || !(cfg->sigma[d] >= 0.0f))
            return fail(BN_ERR_INVALID, "need u_min <= u_max and sigma >= 0");
    // The default arithmetic carries the heading vector by a small-angle rotation per step (rotate_spec, bn_device_math.h): its
    // degree-6 polynomials are good to 1e-7 up to 0.5 rad per step (the reference's dt = 0.1, |omega| <= 1 give 0.1), and the
    // branch-free near form of the heading wrap needs less than pi.  The reference itself has no such bound (transit takes any
    // delta_t, robot_model.py:60): beyond it the handle takes the reference-order arithmetic -- sincos_spec of every step's heading,
    // general wrap -- exactly as if BN_FLAG_REFERENCE_ORDER had been given (bn_mppi_arithmetic() tells).
    const bool big_step = !((double)cfg->dt * std::max(std::fabs((double)cfg->u_min[1]), std::fabs((double)cfg->u_max[1])) <= 0.5);
    if (big_step && !std::isfinite((double)cfg->dt * ((double)cfg->u_max[1] - (double)cfg->u_min[1])))
        return fail(BN_ERR_INVALID, "dt and the angular-velocity bounds must be finite");
    // The kernels share one gather between stage cost t and transit t+1; that needs the upper clamp
    // to land in the last cell, as it does for every reference GridMap (grid_map.py:42-50).
    const float span_x = (cfg->x_limits[1] - cfg->x_limits[0]) / cfg->resolution;
    const float span_y = (cfg->y_limits[1] - cfg->y_limits[0]) / cfg->resolution;
    if (!(span_x >= (float)(cfg->grid_size - 1)) || !(span_y >= (float)(cfg->grid_size - 1)))
        return fail(BN_ERR_INVALID, "x/y_limits must span at least grid_size-1 cells of `resolution`");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BN_ERR_NO_DEVICE, "no HIP device visible: the MPPI planner has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(BN_ERR_INVALID, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    DeviceGuard guard(cfg->device_id);
    if (!guard.ok) return fail(BN_ERR_HIP, "hipSetDevice(%d) failed", cfg->device_id);
    hipDeviceProp_t prop;
    BN_HIP(hipGetDeviceProperties(&prop, cfg->device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BN_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", cfg->device_id,
                    prop.gcnArchName);

    bn_mppi *h = new bn_mppi();
    h->cfg = *cfg;
    bn::SolveParams &p = h->p;
    p.K = cfg->num_samples; p.T = cfg->horizon; p.G = cfg->grid_size; p.B = cfg->num_instances;
    p.nblk = (p.K + bn::kRolloutsPerBlock - 1) / bn::kRolloutsPerBlock;
    p.Kp = p.nblk * bn::kRolloutsPerBlock;
    p.res = cfg->resolution; p.inv_res = 1.0f / cfg->resolution;
    p.pow2 = is_pow2_float(cfg->resolution) ? 1 : 0;
    p.x0 = cfg->x_limits[0]; p.y0 = cfg->y_limits[0];
    p.x_hi = cfg->x_limits[1]; p.y_hi = cfg->y_limits[1];
    p.dt = cfg->dt; p.thr = cfg->stuck_threshold; p.lambda_ = cfg->lambda_;
    {
        int e = 0;
        const float m = std::frexp(cfg->lambda_, &e);                  // lambda_ = m * 2^e, m in [0.5, 1)
        p.inv_lambda = (m == 0.5f && e > -120 && e < 120) ? std::ldexp(1.0f, 1 - e) : 0.0f;
    }
    p.sigma0 = cfg->sigma[0]; p.sigma1 = cfg->sigma[1];
    p.iv0 = cfg->inv_var[0]; p.iv1 = cfg->inv_var[1];
    p.umin0 = cfg->u_min[0]; p.umax0 = cfg->u_max[0];
    p.umin1 = cfg->u_min[1]; p.umax1 = cfg->u_max[1];
    p.seed = cfg->seed;
    p.store_u = (cfg->flags & BN_FLAG_STORE_CONTROLS) ? 1 : 0;
    p.lean = (cfg->flags & BN_FLAG_LEAN) ? 1 : 0;
    // One-wave kernel, library noise: how much of the horizon the epilogue draws again instead of reading the controls back (see
    // rollout_wave.inc).  Everything with the trajectory dump (its stores already load the memory system), nothing in lean mode.
    p.regen_steps = p.T;
    p.park_steps = bn::kWaveParkSteps;
    if (const char *e = exp_env("BN_REGEN_STEPS")) p.regen_steps = std::max(0, atoi(e));      // experiments: tools/wave_ab.py
    if (const char *e = exp_env("BN_PARK_STEPS")) p.park_steps = std::min(std::max(0, atoi(e)), bn::kWaveParkSteps);
    p.lds_park = 0;                                    // decided below, when the window is known
    p.ref_order = ((cfg->flags & BN_FLAG_REFERENCE_ORDER) || big_step) ? 1 : 0;
    p.wrap_near = ((double)cfg->dt * std::max(std::fabs((double)cfg->u_min[1]), std::fabs((double)cfg->u_max[1])) < 3.0) ? 1 : 0;
    // Workgroup i of a launch runs on XCD i % 8 (observed, used for speed only).  xs = 3 interleaves 8 instances along grid x
    // so that the workgroups of one instance share an XCD's L2 (rollout_grid); measured SLOWER (64 instances: 29.2 vs 28.1 us,
    // 60: 28.5 vs 24.9): the dispatcher then fills the CUs unevenly (3 to 5 workgroups per CU instead of 4, tools/block_trace.py)
    // and a few workgroups wait for a second round.  Default: instance per grid row, workgroups of an instance spread over the XCDs.
    p.xs = 0;
    if (const char *e = exp_env("BN_XCD_PACK")) p.xs = (e[0] == '1') ? 3 : 0;      // experiments (tools/r2_measure.py)

    h->n_maps = (cfg->flags & BN_FLAG_SHARED_MAP) ? 1 : p.B;
    p.map_stride = (h->n_maps == 1) ? 0 : p.G * p.G;

    // Reachable window: |dx| per step <= trav*|v|*dt <= vmax*dt (robot_model.py:82,86-87).
    const double vmax = std::max(std::fabs((double)cfg->u_min[0]), std::fabs((double)cfg->u_max[0]));
    const double reach_cells = std::ceil((double)p.T * vmax * (double)cfg->dt / (double)cfg->resolution) + 1.0;
    p.reach = (int)std::min(reach_cells, (double)p.G);
    // reach either side of the start cell (reach has a cell to spare), up to the whole map plus the guard row / column at index G
    // (window_origin_wide).  An odd pitch on purpose: with 2 reach + 2 vertically adjacent cells collided in the LDS banks more
    // often (K=16384 T=100: 27.4 -> 28.5 us per solve).
    p.WN = std::min(p.G + 1, 2 * p.reach + 1);
    const size_t lds_budget = 160 * 1024;
    // The clamp-free window gather (trav_window) relies on the raw cell of an in-limits position being <= G, the guard row / column:
    // true for every reference GridMap (limits span exactly G cells, grid_map.py:42-50).  Limits that span MORE cells than the grid
    // has put raw cells beyond the guard for positions clamped to the upper limit; such a geometry takes the clamped gather from
    // global memory (the reference's index clamp, grid_map.py:209), like BN_FLAG_NO_LDS_WINDOW.
    const bool wide_limits = span_x > (float)cfg->grid_size * (1.0f + 1e-6f) || span_y > (float)cfg->grid_size * (1.0f + 1e-6f);
    if ((cfg->flags & BN_FLAG_NO_LDS_WINDOW) || wide_limits) p.WN = 0;
    // Slow path: a horizon whose control tile does not fit the role kernels' LDS (MPPI.__init__ takes any horizon, mppi.py:25), or
    // the reference-order arithmetic in sampled-slip mode (its fused kernel has no such variant).  The one-wave kernel keeps its
    // controls in HBM and has no tile: rollouts + stand-alone tail, two launches per solve, one stream.  (The reference order by
    // itself runs on every kernel: rollout_role_ref_*.hip, rollout_wave_ref.hip.)
    h->slow_path = p.ref_order != 0 && (cfg->flags & BN_FLAG_SAMPLED_SLIP) != 0;
    if (!h->slow_path) {
        if (bn::rollout_lds_bytes(p) > lds_budget) p.WN = 0;
        if (bn::rollout_lds_bytes(p) > lds_budget) { h->slow_path = true; p.WN = wide_limits || (cfg->flags & BN_FLAG_NO_LDS_WINDOW) ? 0 : std::min(p.G + 1, 2 * p.reach + 1); }
    }
    // Sampled slip: the tail's optimal rollout stages a (mean, std) window on top of the plain one -- three windows' worth of LDS.  A
    // reach that fits the rollout kernels (which fall back to the global gather by themselves, sampled_fused) can still be too much
    // for the tail; then everything gathers from global memory.  (Round 4: tools/fuzz_sweep.py, seed 2327 -- T = 63, 0.1 m cells,
    // 1 m/s: a 129 x 129 window, 200 KB in the tail, and the launch failed with "invalid argument".)
    if ((cfg->flags & BN_FLAG_SAMPLED_SLIP) && !h->slow_path && bn::finish_lds_bytes_for(p, true) > lds_budget) p.WN = 0;
    if (h->slow_path) {
        const bool sampled = (cfg->flags & BN_FLAG_SAMPLED_SLIP) != 0;
        if (bn::wave_lds_bytes(p) > lds_budget || (sampled && bn::finish_lds_bytes_for(p, true) > lds_budget)) p.WN = 0;
        const size_t need = sampled ? sizeof(float) * (4 * (size_t)p.T + 2 * (size_t)p.T * bn::kUPad + 64) : bn::wave_lds_bytes(p);
        if (need > lds_budget || bn::finish_lds_bytes_for(p, sampled) > lds_budget) {
            delete h;
            return fail(BN_ERR_INVALID, "horizon %d needs %zu B of LDS even on the slow path (> 160 KiB)", p.T, std::max(need, bn::finish_lds_bytes_for(p, sampled)));
        }
    }

    {
        const size_t by_lds = lds_budget / std::max<size_t>(bn::rollout_lds_bytes(p), 1);
        const size_t by_waves = 32 / (bn::kRolloutThreads / 64);
        h->resident_wgs = std::max<size_t>(1, std::min(by_lds, by_waves)) * (size_t)std::max(prop.multiProcessorCount, 1);
        h->n_cus = std::max(prop.multiProcessorCount, 1);
    }
    int rc = BN_OK;
    auto alloc = [&](auto **ptr, size_t bytes) {
        if (rc != BN_OK) return;
        hipError_t e = hipMalloc((void **)ptr, bytes);
        if (e == hipSuccess) e = hipMemset(*ptr, 0, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();       // reported through rc: the sticky error must not surface in the caller's next torch call
            rc = fail(BN_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
    };
    const size_t B = p.B, K = p.K, T = p.T, G = p.G;
    alloc(&h->d_map, (size_t)h->n_maps * G * G * 4);
    alloc(&h->d_state, B * 3 * 4);
    alloc(&h->d_goal, B * 2 * 4);
    alloc(&h->d_mean, B * T * 2 * 4);
    if (!p.lean) alloc(&h->d_X, B * (T + 1) * 3 * (size_t)p.Kp * 4);      // lean mode never materialises _state_seq_batch
    alloc(&h->d_mean_used, B * T * 2 * 4);
    // the throughput kernel keeps its controls in this buffer instead of an LDS tile, requested or not
    const bool want_wave = !h->slow_path && !(cfg->flags & (BN_FLAG_NO_PIPELINE | BN_FLAG_ROLE_KERNEL | BN_FLAG_SAMPLED_SLIP)) && p.nblk <= 32 &&
                           ((cfg->flags & BN_FLAG_WAVE_KERNEL) ||
                            (((size_t)p.B + 1) * (p.nblk + 1) > (size_t)h->n_cus && wave_kernel_is_faster(p, h->resident_wgs, h->n_cus)));
    if (p.store_u || want_wave || h->slow_path) alloc(&h->d_U, B * T * 2 * (size_t)p.Kp * 4);
    // one-wave kernel: a second control tile in LDS for the chunk behind the register block, if 16 workgroups per CU still fit
    if (want_wave && p.park_steps > 0 && p.regen_steps > 0 && p.WN > 0 && !exp_env("BN_NO_LDS_PARK")) {
        p.lds_park = 1;
        if (bn::wave_lds_bytes(p) > (size_t)160 * 1024 / 16) p.lds_park = 0;
    }
    for (int q = 0; q < kSlots; ++q) {
        alloc(&h->d_cost[q], B * K * 4);
        alloc(&h->d_part[q], B * (size_t)p.nblk * (2 + 2 * T) * 4);
        alloc(&h->d_state_copy[q], B * 3 * 4);
    }
    alloc(&h->d_cost_out, B * K * 4);
    alloc(&h->d_w, B * K * 4);
    alloc(&h->d_ustar, (B * T * 2 + B * (T + 1) * 3) * 4);   // U* then X* in ONE block (BN_BUF_USTAR_XSTAR): a caller copies both at once
    if (h->d_ustar) h->d_xstar = h->d_ustar + B * T * 2;
    alloc(&h->d_stats, B * 2 * 4);
    if (rc == BN_OK && hipHostMalloc((void **)&h->h_pinned, B * 3 * 4, hipHostMallocDefault) != hipSuccess)
        rc = fail(BN_ERR_HIP, "hipHostMalloc failed");
    if (rc == BN_OK) {
        if (!(cfg->flags & BN_FLAG_PRIVATE_STREAM)) {
            h->stream = (hipStream_t)cfg->stream;          // may be the null stream
        } else if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess) {
            h->own_stream = true;
        } else {
            rc = fail(BN_ERR_HIP, "hipStreamCreate failed");
        }
    }
    if (rc != BN_OK) {
        bn_mppi_destroy(h);
        return rc;
    }
    p.map = h->d_map; p.state = h->d_state; p.goal = h->d_goal; p.mean = h->d_mean; p.eps = nullptr;
    p.X = h->d_X; p.U = h->d_U; p.cost = h->d_cost[0]; p.part = h->d_part[0]; p.w = h->d_w;
    p.state_copy = h->d_state_copy[0]; p.cost_out = h->d_cost_out;
    p.ustar = h->d_ustar; p.xstar = h->d_xstar; p.stats = h->d_stats; p.mean_used = h->d_mean_used;
    // every rollout workgroup re-merges the previous solve's nblk partials: only worth it while they are few
    p.slip_on = (cfg->flags & BN_FLAG_SAMPLED_SLIP) ? 1 : 0;
    // (the fused sampled-slip kernel has its own, larger LDS layout: what an overlapped batch may count on being resident -- the
    // residency rules of bn_mppi_solve_n_async -- is that kernel's figure, not the role kernel's)
    if (p.slip_on && bn::sampled_fused(p)) h->resident_wgs = bn::sampled_resident_per_cu(p) * (size_t)h->n_cus;
    h->pipelined = !h->slow_path && !(cfg->flags & BN_FLAG_NO_PIPELINE) && p.nblk <= kPipelinedMaxBlocks && !p.slip_on;
    // throughput kernel: launches with more workgroups than the role kernel keeps resident in one round (4 per CU)
    h->wave_kernel = h->pipelined && want_wave;
    // latency variant: every workgroup (rollouts + aux) alone on a CU -- with room left for one instance of an overlapped successor
    // (15 instances of K=1024 fill 255 of 256 CUs: 15.3 us per launch against the role kernel's 13.9; 14 instances 13.1 against
    // 13.9) --, its LDS layout must fit, not forced elsewhere
    h->lat_kernel = h->pipelined && !h->wave_kernel && !(cfg->flags & BN_FLAG_ROLE_KERNEL) && bn::lat_lds_bytes(p) > 0 &&
                    ((cfg->flags & BN_FLAG_LAT_KERNEL) || ((size_t)p.B + 1) * (p.nblk + 1) <= (size_t)std::max(prop.multiProcessorCount, 1));
    if (const char *e = exp_env("BN_LAT_KERNEL")) h->lat_kernel = h->lat_kernel && e[0] != '0';      // experiments
    alloc(&h->d_flags, ((kSlots + 1) * B + 3) * bn::kFlagStride * sizeof(unsigned long long));      // [kSlots][B] + [B] counters, a spare slot, the device error word
    alloc(&h->d_mean_snap, B * T * 2 * 4);
    // (the journal's state snapshots, d_state_snaps, are allocated below, and only for a handle that can overlap at all)
    if (rc == BN_OK) {
        if (hipHostMalloc((void **)&h->h_err, 64, hipHostMallocMapped) != hipSuccess) rc = fail(BN_ERR_HIP, "hipHostMalloc (error word) failed");
        else if (hipHostGetDevicePointer((void **)&h->d_err, h->h_err, 0) != hipSuccess) rc = fail(BN_ERR_HIP, "hipHostGetDevicePointer failed");
        else *h->h_err = 0;
    }
    if (rc == BN_OK) {
        if (hipHostMalloc((void **)&h->h_mail, B * 2 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&h->d_mail, h->h_mail, 0) != hipSuccess) rc = fail(BN_ERR_HIP, "hipHostMalloc (first-action mailbox) failed");
        else std::memset(h->h_mail, 0, B * 2 * sizeof(unsigned long long));
    }
    p.mail = h->d_mail;
    if (h->lat_kernel && p.nblk <= 16 && 2 * p.T <= bn::kRolloutThreads && !exp_env("BN_NO_GRANULES"))
        for (int q = 0; q < kSlots; ++q) alloc(&h->d_gran[q], (B * (size_t)p.nblk * (2 + 2 * T) + 4 * B) * sizeof(unsigned long long));   // rows, then 4 per instance for the state
    // the role kernel (launches that do not leave every workgroup a CU of its own) overlaps its launches as well: a workgroup of
    // the next solve takes the slot a finished one frees and waits there for ITS instance's previous solve only
    h->role_overlap = h->pipelined && !h->lat_kernel && !(cfg->flags & BN_FLAG_NO_OVERLAP);
    if (const char *e = exp_env("BN_ROLE_OVERLAP")) h->role_overlap = h->role_overlap && e[0] != '0';   // experiments
    // the deterministic kernel's ticket path (K > 4096: one launch per solve, merge by the last workgroup) overlaps its launches as well
    const bool ticket_det = !h->slow_path && !p.slip_on && !h->pipelined && !(cfg->flags & BN_FLAG_NO_PIPELINE) && p.nblk <= 1024 &&
                            bn::finish_lds_bytes(p) + 256 <= bn::rollout_lds_bytes(p);
    h->ticket_overlap = (ticket_det || (!(cfg->flags & BN_FLAG_NO_PIPELINE) && bn::sampled_fused(p))) && !(cfg->flags & BN_FLAG_NO_OVERLAP) &&
                        !exp_env("BN_NO_TICKET_OVERLAP");
    const bool may_overlap = (h->lat_kernel || h->role_overlap || h->ticket_overlap) && !(cfg->flags & BN_FLAG_NO_OVERLAP);
    if (may_overlap) {
        // Two launches in flight.  Measured with three (role kernel, 64 instances): 23.3 instead of 22.6 us per launch; with
        // four the launches starve each other of slots (waits expire).  The slot / buffer arithmetic below holds for up to kMaxStreams.
        h->n_streams = 2;
        if (const char *e = exp_env("BN_OVERLAP_STREAMS")) h->n_streams = std::min(std::max(std::atoi(e), 2), kMaxStreams);   // experiments
        for (int q = 0; q + 1 < h->n_streams; ++q) {
            if (h->d_X) alloc(&h->d_Xalt[q], B * (T + 1) * 3 * (size_t)p.Kp * 4);
            if (h->d_U) alloc(&h->d_Ualt[q], B * T * 2 * (size_t)p.Kp * 4);
        }
        // the journal's state snapshots: only a handle that can overlap ever journals (up to 16 MB per handle otherwise,
        // times the planners of a K-sharded or per-instance set-up).  At most 16 MB; B <= 341 get the journal's own 4096 entries.
        h->snap_cap = std::max<size_t>(64, std::min<size_t>(4096, ((size_t)16 << 20) / ((size_t)B * 12)));
        alloc(&h->d_state_snaps, h->snap_cap * B * 3 * 4);
    }
    if (rc == BN_OK && may_overlap) {
        bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; ok && q < h->n_streams; ++q) ok = hipEventCreateWithFlags(&h->ev_guard[q], hipEventDisableTiming) == hipSuccess;
        for (int q = 0; ok && q + 1 < h->n_streams; ++q)
            ok = hipStreamCreateWithFlags(&h->xstream[q], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&h->ev_join[q], hipEventDisableTiming) == hipSuccess;
        if (!ok) rc = fail(BN_ERR_HIP, "stream / event creation for overlapped launches failed");
    }
    // (late allocations also go through `alloc`: a failure anywhere destroys the handle and everything it owns)
    if (p.slip_on) {
        alloc(&h->d_slip_std, (size_t)h->n_maps * p.G * p.G * 4);
        p.slip_std = h->d_slip_std;
        for (int q = 0; q < kSlots; ++q) {
            alloc(&h->d_ustar2[q], (size_t)p.B * p.T * 2 * 4);
            alloc(&h->d_stats2[q], (size_t)p.B * 2 * 4);
        }
        alloc(&h->d_ticket, (size_t)p.B * 65 * 4);
        alloc(&h->d_gpart, (size_t)p.B * 64 * (2 + 2 * p.T) * 4);
        p.ticket = h->d_ticket; p.gpart = h->d_gpart;
        h->ticket_mode = !h->slow_path && !(cfg->flags & BN_FLAG_NO_PIPELINE) && bn::sampled_fused(p);
    } else if (ticket_det) {
        // K > 4096: too many partials for every workgroup to re-merge; the last workgroup of a launch merges them
        for (int q = 0; q < kSlots; ++q) {
            alloc(&h->d_ustar2[q], (size_t)p.B * p.T * 2 * 4);
            alloc(&h->d_stats2[q], (size_t)p.B * 2 * 4);
        }
        alloc(&h->d_ticket, (size_t)p.B * 65 * 4);
        alloc(&h->d_gpart, (size_t)p.B * 64 * (2 + 2 * p.T) * 4);
        p.gpart = h->d_gpart;
        h->ticket_mode = true;             // p.ticket stays null in h->p: only the one-launch path selects the ticket kernel
    }
    if (rc == BN_OK && !p.pow2) {
        // general resolution: the in-loop lookups divide with the three-instruction quotient (quotient_general); check it over every
        // float they can see -- ~1e9 values, a millisecond -- before anything relies on it
        unsigned long long *bad = h->d_flags + ((kSlots + 1) * B + 1) * bn::kFlagStride;      // the spare counter slot, zero
        const float d_max = std::max(p.x_hi - p.x0, p.y_hi - p.y0);
        unsigned long long n_bad = 1;
        if (!(std::isfinite(d_max) && d_max >= 0.0f) || bn::launch_quotient_check(p.res, p.inv_res, d_max, bad, nullptr) != hipSuccess ||
            hipMemcpy(&n_bad, bad, sizeof n_bad, hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(BN_ERR_HIP, "checking the cell-index quotient for resolution %g failed", (double)p.res);
        } else if (n_bad != 0) {
            rc = fail(BN_ERR_INVALID, "resolution %.9g: the correctly rounded three-instruction quotient disagrees with the division for %llu "
                                      "positions in [0, %g] (no such resolution was known: please report it); use a neighbouring value",
                      (double)p.res, n_bad, (double)d_max);
        } else {
            p.fast_div = 1;
        }
        (void)hipMemset(bad, 0, sizeof n_bad);
    }
    if (rc == BN_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(BN_ERR_HIP, "hipDeviceSynchronize failed after allocation");
    // Overlapped launches need an extra stream that DISPATCHES concurrently with the handle's stream.  HIP deals its streams onto a
    // handful of hardware queues in creation order, so which queue a fresh stream lands on depends on how many streams the process
    // has created before: a process that brought up RCCL first got the handle's own queue (15.2 instead of 9.6 us per solve: no
    // overlap at all).  Ask the hardware (launch_queue_probe: can the candidate dispatch while a grid larger than the chip is still
    // being placed on the handle's stream?); a candidate that fails is parked and a fresh stream takes its place.
    if (rc == BN_OK && may_overlap) {
        int *probe = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 1) * bn::kFlagStride);      // the spare counter slot
        for (int q = 0; rc == BN_OK && q + 1 < h->n_streams; ++q) {
            bool found = false;
            for (int attempt = 0; attempt < 8; ++attempt) {
                int seen = 0;
                if (hipMemset(probe, 0, 2 * sizeof(int)) != hipSuccess || bn::launch_queue_probe(probe, h->stream, h->xstream[q], h->n_cus) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess || hipStreamSynchronize(h->xstream[q]) != hipSuccess ||
                    hipMemcpy(&seen, probe + 1, sizeof seen, hipMemcpyDeviceToHost) != hipSuccess) {
                    (void)hipGetLastError();
                    rc = fail(BN_ERR_HIP, "probing the streams of overlapped launches failed");
                    break;
                }
                if (seen) { found = true; break; }
                if (attempt == 7) break;                   // (the stream in hand has been probed and failed: do not swap it for an unprobed one)
                hipStream_t fresh = nullptr;
                if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
                h->parked.push_back(h->xstream[q]);        // kept alive until destroy: destroying it would hand its queue slot to the next one
                h->xstream[q] = fresh;
            }
            if (exp_env("BN_DEBUG_CREATE")) std::fprintf(stderr, "[bn_mppi_create] extra stream %d: %s after %zu replacement(s)\n", q, found ? "dispatches concurrently" : "NO concurrent queue found", h->parked.size());
            // No stream that dispatches concurrently with the handle's: launches of a batch would not overlap, and workgroups waiting
            // on the device for a predecessor the queue has not started yet would only burn their bounded waits.  One stream then.
            if (!found && rc == BN_OK) h->overlap_off = true;
        }
        (void)hipMemset(probe, 0, 2 * sizeof(int));
    }
    // Host-paced loop (BN_FLAG_HOST_PACED): a second private stream that dispatches concurrently with the first, the request words, events
    if (rc == BN_OK && (cfg->flags & BN_FLAG_HOST_PACED) && p.B == 1 && h->lat_kernel && h->d_gran[0] && may_overlap && !h->overlap_off &&
        h->n_streams > 1 && h->xstream[0] && !p.slip_on && !(cfg->flags & (BN_FLAG_NO_PIPELINE | BN_FLAG_PROFILE))) {
        {
            bn::SolveParams q = p;
            q.spec_extra = h->hp_extra;
            if (bn::lat_lds_bytes(q) == 0) h->hp_extra = 0;        // no room for the wider window: every state but the previous one stages again
        }
        int *probe = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 1) * bn::kFlagStride);      // the spare counter slot
        bool found = false, ok = hipStreamCreateWithFlags(&h->xstream[1], hipStreamNonBlocking) == hipSuccess;
        for (int attempt = 0; ok && attempt < 8 && !found; ++attempt) {
            int seen = 0;
            ok = hipMemset(probe, 0, 2 * sizeof(int)) == hipSuccess && bn::launch_queue_probe(probe, h->xstream[0], h->xstream[1], h->n_cus) == hipSuccess &&
                 hipStreamSynchronize(h->xstream[0]) == hipSuccess && hipStreamSynchronize(h->xstream[1]) == hipSuccess &&
                 hipMemcpy(&seen, probe + 1, sizeof seen, hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) break;
            if (seen) { found = true; break; }
            if (attempt == 7) break;
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
            h->parked.push_back(h->xstream[1]);
            h->xstream[1] = fresh;
        }
        (void)hipGetLastError();
        (void)hipMemset(probe, 0, 2 * sizeof(int));
        if (ok && found) {
            const size_t words = (size_t)kSlots * 8 + 16;      // request slots | [0] spare, [1..4] acknowledgements, [7] create's echo | [8..11] "gave up" per slot
            ok = hipHostMalloc((void **)&h->h_req, words * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess &&
                 hipHostGetDevicePointer((void **)&h->d_req, h->h_req, 0) == hipSuccess;
            if (ok) std::memset(h->h_req, 0, words * sizeof(unsigned long long));
            alloc(&h->d_req_dev, (size_t)kSlots * 8 * sizeof(unsigned long long));
            for (int q = 0; ok && q < 2; ++q) ok = hipEventCreateWithFlags(&h->hp_ev[q], hipEventDisableTiming) == hipSuccess;
            if (ok && rc == BN_OK) {
                h->hp_stream[0] = h->xstream[0]; h->hp_stream[1] = h->xstream[1];
                h->hp_enabled = true;
                // request words in device memory the host can write (see bar_alloc) -- taken only if a kernel reads back what the CPU wrote
                if (!std::getenv("BENCHNAV_NO_BAR_REQUEST")) {
                    unsigned long long *bar = static_cast<unsigned long long *>(bar_alloc(cfg->device_id, words * sizeof(unsigned long long)));
                    if (bar) {
                        bool good = true;
                        for (size_t i = 0; i < words; ++i) bar[i] = 0;
                        const unsigned long long magic = 0x5a5a0000c3c30001ull;
                        __atomic_store_n(bar + 3, magic, __ATOMIC_RELEASE);
                        __builtin_ia32_sfence();
                        good = bn::launch_echo64(bar + 3, h->d_req + (size_t)kSlots * 8 + 7, h->xstream[0]) == hipSuccess && hipStreamSynchronize(h->xstream[0]) == hipSuccess &&
                               __atomic_load_n(h->h_req + (size_t)kSlots * 8 + 7, __ATOMIC_ACQUIRE) == magic;
                        (void)hipGetLastError();
                        __atomic_store_n(bar + 3, 0ull, __ATOMIC_RELEASE);
                        __builtin_ia32_sfence();
                        h->h_req[(size_t)kSlots * 8 + 7] = 0;
                        if (good) h->req_bar = bar; else bar_free(bar);
                    }
                }
            }
        }
        (void)hipGetLastError();                       // (a handle that cannot pace keeps the one-launch path: bn_mppi_host_paced() tells)
    }
    if (rc != BN_OK) {
        bn_mppi_destroy(h);
        return rc;
    }
    {
        std::lock_guard<std::mutex> lock(g_overlap_mu);
        g_handles[h->cfg.device_id].push_back(h);
    }
    *out = h;
    return BN_OK;
}

static void shard_comm_release(bn_mppi *h);      // (with the RCCL loader, further down)

void bn_mppi_destroy(bn_mppi_t *h)
{
    if (!h) return;
    DeviceGuard guard(h->cfg.device_id);
    hp_cancel(h);
    (void)hipStreamSynchronize(h->stream);
    {
        std::lock_guard<std::mutex> lock(g_overlap_mu);
        auto it = g_overlap_owner.find(h->cfg.device_id);
        if (it != g_overlap_owner.end() && it->second == h) g_overlap_owner.erase(it);
        g_overlap_owners.store((int)g_overlap_owner.size(), std::memory_order_relaxed);
        auto &v = g_handles[h->cfg.device_id];
        v.erase(std::remove(v.begin(), v.end(), h), v.end());
    }
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->shard_comm || h->shard_prepared) shard_comm_release(h);
    void *bufs[] = {h->d_map, h->d_state, h->d_goal, h->d_mean, h->d_eps, h->d_X, h->d_U, h->d_cost_out,
                    h->d_w, h->d_ustar /* d_xstar lives in the same block */, h->d_stats, h->d_scratch, h->d_idx, h->d_lat_mean, h->d_lat_std,
                    h->d_ep_states, h->d_ep_reward, h->d_env_state, h->d_ep_done, h->d_ep_action, h->d_slip_std,
                    h->d_ustar2[0], h->d_ustar2[1], h->d_ustar2[2], h->d_ustar2[3], h->d_stats2[0], h->d_stats2[1], h->d_stats2[2], h->d_stats2[3],
                    h->d_ticket, h->d_gpart, h->d_mean_used};
    static_assert(kSlots == 4, "the list above names the four slots");
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (h->d_flags) (void)hipFree(h->d_flags);
    for (int q = 0; q < kSlots; ++q) {
        if (h->d_gran[q]) (void)hipFree(h->d_gran[q]);
        if (h->d_cost[q]) (void)hipFree(h->d_cost[q]);
        if (h->d_part[q]) (void)hipFree(h->d_part[q]);
        if (h->d_state_copy[q]) (void)hipFree(h->d_state_copy[q]);
    }
    for (int q = 0; q < kMaxStreams - 1; ++q) {
        if (h->d_Xalt[q]) (void)hipFree(h->d_Xalt[q]);
        if (h->d_Ualt[q]) (void)hipFree(h->d_Ualt[q]);
        if (h->xstream[q]) { (void)hipStreamSynchronize(h->xstream[q]); (void)hipStreamDestroy(h->xstream[q]); }
        if (h->ev_join[q]) (void)hipEventDestroy(h->ev_join[q]);
    }
    for (hipStream_t ps : h->parked) (void)hipStreamDestroy(ps);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_guard) if (e) (void)hipEventDestroy(e);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->h_err) (void)hipHostFree(h->h_err);
    if (h->h_mail) (void)hipHostFree(h->h_mail);
    if (h->h_req) (void)hipHostFree(h->h_req);
    if (h->req_bar) bar_free(h->req_bar);
    if (h->d_req_dev) (void)hipFree(h->d_req_dev);
    for (hipEvent_t e : h->hp_ev) if (e) (void)hipEventDestroy(e);
    if (h->d_mean_snap) (void)hipFree(h->d_mean_snap);
    if (h->d_state_snaps) (void)hipFree(h->d_state_snaps);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int bn_mppi_set_map(bn_mppi_t *h, int32_t instance, const float *risk, bn_mem_kind where)
{
    if (int rc = check_instance(h, instance, true)) return rc;
    if (!risk) return fail(BN_ERR_INVALID, "risk is null");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t bytes = (size_t)h->p.G * h->p.G * 4;
    const hipMemcpyKind kind = where == BN_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const int lo = instance < 0 ? 0 : std::min(instance, h->n_maps - 1);
    const int hi = instance < 0 ? h->n_maps : lo + 1;
    if (int rc = flush_tail(h)) return rc;           // the pending tail rolls X* out on the current map
    BN_HIP(hipStreamSynchronize(h->stream));
    for (int m = lo; m < hi; ++m) BN_HIP(hipMemcpy(h->d_map + (size_t)m * h->p.G * h->p.G, risk, bytes, kind));
    h->map_set = true;
    h->map_epoch += 1;
    return BN_OK;
}

int bn_mppi_set_slip_std(bn_mppi_t *h, int32_t instance, const float *stdv, bn_mem_kind where)
{
    if (int rc = check_instance(h, instance, true)) return rc;
    if (!stdv) return fail(BN_ERR_INVALID, "std is null");
    if (!h->p.slip_on) return fail(BN_ERR_STATE, "handle was created without BN_FLAG_SAMPLED_SLIP");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t bytes = (size_t)h->p.G * h->p.G * 4;
    const hipMemcpyKind kind = where == BN_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const int lo = instance < 0 ? 0 : std::min(instance, h->n_maps - 1), hi = instance < 0 ? h->n_maps : lo + 1;
    for (int m = lo; m < hi; ++m) BN_HIP(hipMemcpy(h->d_slip_std + (size_t)m * h->p.G * h->p.G, stdv, bytes, kind));
    h->slip_std_set = true;
    return BN_OK;
}

int bn_mppi_set_slip_noise(bn_mppi_t *h, const float *zt_device, const float *zc_device, const float *zo_device)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->p.slip_on) return fail(BN_ERR_STATE, "handle was created without BN_FLAG_SAMPLED_SLIP");
    if ((zt_device == nullptr) != (zc_device == nullptr) || (zt_device == nullptr) != (zo_device == nullptr))
        return fail(BN_ERR_INVALID, "give all three arrays or none");
    h->p.zt = zt_device; h->p.zc = zc_device; h->p.zo = zo_device;
    return BN_OK;
}

int bn_mppi_set_goal(bn_mppi_t *h, int32_t instance, const float goal_host[2])
{
    if (int rc = check_instance(h, instance, true)) return rc;
    if (!goal_host) return fail(BN_ERR_INVALID, "goal is null");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const int lo = instance < 0 ? 0 : instance, hi = instance < 0 ? h->p.B : instance + 1;
    for (int b = lo; b < hi; ++b) BN_HIP(hipMemcpy(h->d_goal + b * 2, goal_host, 8, hipMemcpyHostToDevice));
    h->goal_set = true;
    return BN_OK;
}

int bn_mppi_set_mean(bn_mppi_t *h, int32_t instance, const float *mean_host)
{
    if (int rc = check_instance(h, instance, true)) return rc;
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;           // afterwards the mean buffer is authoritative again
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->p.T * 2;
    const int lo = instance < 0 ? 0 : instance, hi = instance < 0 ? h->p.B : instance + 1;
    for (int b = lo; b < hi; ++b) {
        if (mean_host) BN_HIP(hipMemcpy(h->d_mean + b * n, mean_host, n * 4, hipMemcpyHostToDevice));
        else BN_HIP(hipMemset(h->d_mean + b * n, 0, n * 4));
    }
    return BN_OK;
}

int bn_mppi_get_mean(bn_mppi_t *h, int32_t instance, float *mean_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!mean_host) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->p.T * 2;
    BN_HIP(hipMemcpy(mean_host, h->d_mean + instance * n, n * 4, hipMemcpyDeviceToHost));
    return BN_OK;
}

// A launch of handle h that is not a member of an overlapped batch of its own: if ANOTHER handle's overlapped batch may still be
// in flight on the device, `st` waits for its end (events recorded on that handle's streams).  See g_handles.
static int order_behind_foreign_overlap(bn_mppi *h, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_overlap_mu);
    auto it = g_overlap_owner.find(h->cfg.device_id);
    if (it == g_overlap_owner.end() || it->second == h || !it->second) return BN_OK;
    bn_mppi *o = it->second;
    bool busy = false;
    for (int q = 0; q < o->n_streams; ++q) {
        const hipStream_t os = q ? o->xstream[q - 1] : o->stream;
        if (os == st || (q && !os)) continue;                       // the same stream: ordered anyway
        if (hipStreamQuery(os) == hipSuccess) continue;
        busy = true;
        if (!o->ev_guard[q]) continue;
        BN_HIP(hipEventRecord(o->ev_guard[q], os));
        BN_HIP(hipStreamWaitEvent(st, o->ev_guard[q], 0));
    }
    (void)hipGetLastError();
    if (!busy) { g_overlap_owner.erase(it); g_overlap_owners.store((int)g_overlap_owner.size(), std::memory_order_relaxed); }   // its batch is through: nothing to look at until the next one
    return BN_OK;
}

// Every launch a handle makes on its own stream outside an overlapped batch of its own -- stand-alone tails, re-rolls, environment
// steps and collision checks, not only solves -- goes behind another handle's overlapped batch the same way.
namespace {
int guard_foreign_overlap(bn_mppi *h)
{
    if (h->last_batch_overlapped || g_overlap_owners.load(std::memory_order_relaxed) == 0) return BN_OK;
    return order_behind_foreign_overlap(h, h->stream);
}
}  // namespace

// shard_rollout: the rollouts of a K-sharded solve only (bn_mppi_shard_rollout_async); the tail follows the exchange.
static int solve_impl(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps, bn_noise_kind noise,
                      bool shard_rollout, bool overlap = false, hipStream_t on_stream = nullptr, int alt_buffers = 0, bool self_tail = false)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!states) return fail(BN_ERR_INVALID, "states is null");
    if (!h->map_set || !h->goal_set) return fail(BN_ERR_STATE, "set_map and set_goal must precede solve");
    if ((noise == BN_NOISE_PHILOX) != (eps == nullptr))
        return fail(BN_ERR_INVALID, "eps must be NULL exactly when noise == BN_NOISE_PHILOX");
    BN_BIND(h);
    hp_cancel(h);                                      // (a host-paced launch waiting for a state: this solve is not it)
    h->hp_gran_valid = false;
    if (!overlap && g_overlap_owners.load(std::memory_order_relaxed) > 0) {
        if (int rc = order_behind_foreign_overlap(h, on_stream ? on_stream : h->stream)) return rc;
    }

    bn::SolveParams p = h->p;
    const size_t B = p.B, K = p.K, T = p.T;
    h->map_epoch_at_solve = h->map_epoch;
    // Behind an overlapped batch the host has not checked yet (or batches journalled for a re-run): a plain launch's tail must not
    // write the mean from partials of an expired wait either (flush_tail does the same) -- see raise_wait_expired
    if (h->d_flags && (h->overlap_used || !h->journal.empty()))
        p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 2) * bn::kFlagStride);
    if (states_where == BN_MEM_DEVICE) {
        p.state = states;
    } else {
        // the pinned block may still feed a previous copy: wait for the stream before reusing it
        if (h->solves) BN_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(h->h_pinned, states, B * 3 * 4);
        BN_HIP(hipMemcpyAsync(h->d_state, h->h_pinned, B * 3 * 4, hipMemcpyHostToDevice, h->stream));
        p.state = h->d_state;
    }
    bn::EpsMode mode = bn::kEpsPhilox;
    switch (noise) {
    case BN_NOISE_PHILOX: break;
    case BN_NOISE_HOST_KT2: {
        const size_t bytes = B * K * T * 2 * 4;
        if (bytes > h->eps_bytes) {
            if (h->d_eps) BN_HIP(hipFree(h->d_eps));
            h->d_eps = nullptr; h->eps_bytes = 0;
            BN_HIP(hipMalloc((void **)&h->d_eps, bytes));
            h->eps_bytes = bytes;
        }
        BN_HIP(hipMemcpyAsync(h->d_eps, eps, bytes, hipMemcpyHostToDevice, h->stream));
        p.eps = h->d_eps;
        mode = bn::kEpsKT2;
        break;
    }
    case BN_NOISE_DEVICE_KT2: p.eps = eps; mode = bn::kEpsKT2; break;
    case BN_NOISE_DEVICE_T2K: p.eps = eps; mode = bn::kEpsT2K; break;
    default: return fail(BN_ERR_INVALID, "unknown noise kind %d", (int)noise);
    }

    // Profiling: two-launch mode brackets each kernel with events.  In the pipelined mode a solve is ONE
    // back-to-back launch of ~15 us; an event pair around every launch would add its own ~5 us of
    // dispatch latency, so events are recorded every kProfGroup launches and the mean is taken per group.
    if (shard_rollout) {
        if (int rc = flush_tail(h)) return rc;
        if (h->shard_pending) return fail(BN_ERR_STATE, "the previous sharded solve still waits for bn_mppi_shard_finish_async");
    } else if (h->shard_pending) {
        return fail(BN_ERR_STATE, "a sharded solve waits for bn_mppi_shard_finish_async");
    }
    const bool one_launch = !shard_rollout && (h->pipelined || h->ticket_mode);
    const bool prof_grouped = (h->cfg.flags & BN_FLAG_PROFILE) != 0 && one_launch;
    const bool prof = (h->cfg.flags & BN_FLAG_PROFILE) != 0 && !one_launch;
    hipEvent_t *ev = nullptr;
    if (prof_grouped && h->prof_in_group == 0) {
        if (h->ev_used + 1 > h->ev.size()) {
            hipEvent_t e;
            BN_HIP(hipEventCreate(&e));
            h->ev.push_back(e);
        }
        BN_HIP(hipEventRecord(h->ev[h->ev_used++], h->stream));
    }
    if (prof) {
        if (h->ev_used + 3 > h->ev.size()) {
            for (int i = 0; i < 3; ++i) {
                hipEvent_t e;
                BN_HIP(hipEventCreate(&e));
                h->ev.push_back(e);
            }
        }
        ev = &h->ev[h->ev_used];
        h->ev_used += 3;
        BN_HIP(hipEventRecord(ev[0], h->stream));
    }
    h->last_eps = p.eps;
    h->last_mode = mode;
    const int cur3 = (int)(h->solves % kSlots), prev3 = (int)((h->solves + kSlots - 1) % kSlots);    // per-solve buffers: kSlots slots
    p.solve = h->solves;
    if (h->arm_state_snap) { p.state_snap = h->d_state_snaps + h->snap_slot * (size_t)p.B * 3; h->arm_state_snap = false; }
    p.part = h->d_part[cur3]; p.cost = h->d_cost[cur3]; p.state_copy = h->d_state_copy[cur3];
    p.part_prev = h->d_part[prev3]; p.cost_prev = h->d_cost[prev3]; p.state_prev = h->d_state_copy[prev3];
    if (h->pipelined && !shard_rollout) {
        // one launch: merge + tail of the previous solve ride along with this solve's rollouts
        p.have_prev = h->tail_pending ? 1 : 0;
        p.mean_from_part = h->tail_pending ? 1 : 0;
        p.tail_solve = p.solve - 1;                                   // the tail the aux workgroups write
        if (h->in_episode) {
            // closed loop: from the second step on, the rollout workgroups advance the state themselves
            p.env_on = 1;
            p.closed_loop = (h->ep_len > 0 && h->tail_pending) ? 1 : 0;
            p.ep_index = h->ep_len - 1;                                   // the env step applied / logged by this launch
            p.env_z = (h->ep_z && h->ep_len > 0) ? h->ep_z + (size_t)(h->ep_len - 1) * p.B : nullptr;
            h->ep_len += 1;
        }
        p.wave_kernel = (h->wave_kernel && !h->in_episode) ? 1 : 0;
        if (p.wave_kernel && p.xs == 0 && exp_env("BN_AUX_FIRST")) p.aux_first = 1;      // (experiment builds: VERDICT r5 #3)
        p.lat_kernel = h->lat_kernel ? 1 : 0;
        hipStream_t st = h->stream;
        p.flag_tail = h->d_flags + kSlots * B * bn::kFlagStride;
        if (p.have_prev) {                                             // the aux workgroups write the previous solve's tail
            p.wait_tail = h->tails;                                    // ... after every tail (of the same instance) before it
            h->tails += 1;
        }
        // A synchronous solve on the latency kernel: publishes like a member of a batch (counters, granules) -- for its OWN tail, the second
        // aux workgroup of the same launch; it waits for nobody (no predecessor in flight, no tail pending: the callers see to that).
        const bool lone_self = self_tail && !overlap && h->lat_kernel && !h->in_episode && !p.have_prev && !h->replaying && !h->self_off &&
                               !(h->cfg.flags & BN_FLAG_NO_PIPELINE);
        h->self_tail_launched = false;
        if (lone_self) {
            p.flag_part = h->d_flags;
            p.err = h->d_err; p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 2) * bn::kFlagStride);
            p.cur_slot = cur3; p.prev_slot = prev3;
            p.wait_part = h->pub[prev3];
            p.gran = h->d_gran[cur3];                                  // (null above 16 workgroups per instance: the tail waits for the counter)
            p.overlap = 0;
            h->pub[cur3] += (unsigned long long)p.nblk;
            p.self_tail = 1;
            p.wait_part_self = h->pub[cur3];
            p.wait_tail_self = h->tails;
            h->tails += 1;
            p.out_copy_self = h->self_out_copy;
            if (exp_env("BN_NO_EARLY_MAIL")) p.no_early_mail = 1;      // (experiment builds)
            if (h->inline_state && p.B == 1) { p.state_inline = 1; p.sv0 = h->inline_state[0]; p.sv1 = h->inline_state[1]; p.sv2 = h->inline_state[2]; }
            h->self_used = true;
            h->self_tail_launched = true;
            h->prev_published = false;
            h->hp_gran_valid = p.gran != nullptr;
            h->x_idx = 0;
        } else
        if ((h->lat_kernel || h->role_overlap) && overlap) {   // member of an overlapped batch: publishes, and waits if its predecessor published
            p.flag_part = h->d_flags;
            p.err = h->d_err; p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 2) * bn::kFlagStride);                                          // pinned host memory, mapped
            if (h->arm_snap) { p.mean_snap = h->d_mean_snap; h->arm_snap = false; }
            p.cur_slot = cur3; p.prev_slot = prev3;
            p.wait_part = h->pub[prev3];
            p.gran = p.lat_kernel ? h->d_gran[cur3] : nullptr;
            if (alt_buffers) {                                         // this solve's trajectories / controls go to one of the extra buffers
                if (h->d_Xalt[alt_buffers - 1]) p.X = h->d_Xalt[alt_buffers - 1];
                if (h->d_Ualt[alt_buffers - 1] && p.U) p.U = h->d_Ualt[alt_buffers - 1];
            }
            p.overlap = (h->prev_published && p.have_prev) ? 1 : 0;
            p.gran_prev = (p.overlap && p.lat_kernel) ? h->d_gran[prev3] : nullptr;
            // every member of the batch runs where the round robin put it -- the first one too (no predecessor to wait for, but if it
            // stayed on the handle's stream while the batch is aligned to END there, solves 0 and 1 would share that stream and solve 2,
            // on the other one, would become eligible before solve 1 is even dispatched: for launches that fill the chip that is the
            // starvation pattern described in bn_mppi_solve_n_async -- 64 instances lost the whole gain of overlapping, 27.5 vs 23.3 us)
            if (on_stream) st = on_stream;
            if (p.overlap) h->overlap_used = true;
            h->pub[cur3] += (unsigned long long)p.nblk;
            h->prev_published = true;
            if (self_tail && p.lat_kernel && !h->in_episode) {         // last solve of a batch: its own tail rides in the same launch
                p.self_tail = 1;
                p.wait_part_self = h->pub[cur3];
                p.wait_tail_self = h->tails;
                h->tails += 1;
            }
        } else {
            h->prev_published = false;
        }
        // more workgroups than the role kernel keeps resident at once (4 per CU x 256 CUs): the aux workgroups run in freed slots
        p.aux_prio = (!p.wave_kernel && (size_t)p.B * (p.nblk + 1) > h->resident_wgs) ? 1 : 0;
        if (!p.wave_kernel && !p.store_u) p.U = nullptr;          // the buffer exists for the throughput kernel only
        if (p.self_tail) BN_HIP(bn::launch_rollout_lat_self(p, mode, st));    // (a kernel of its own: rollout_lat.inc, mode 1)
        else BN_HIP(bn::launch_rollout(p, mode, st));
        if (prof_grouped) h->prof_in_group = (h->prof_in_group + 1) % kProfGroup;
        h->solves += 1;
        h->x_idx = 0;                                  // (batches end on the exposed buffers, bn_mppi_solve_n_async)
        h->tail_pending = !p.self_tail;
        if (p.self_tail) h->prev_published = false;                // the next solve starts from the mean that tail writes, in stream order
        if (lone_self) h->last_batch_overlapped = false;
        return BN_OK;
    }
    if (!(h->ticket_overlap && overlap && !shard_rollout)) h->prev_published = false;
    p.have_prev = 0;
    p.mean_from_part = 0;
    p.tail_solve = p.solve;
    if (p.slip_on && !h->slip_std_set) return fail(BN_ERR_STATE, "bn_mppi_set_slip_std must precede solve in sampled-slip mode");
    if (h->ticket_mode && !shard_rollout) {
        // one launch: the rollouts, the ticket merge of this solve, and the previous solve's tail as aux workgroup
        p.have_prev = h->tail_pending ? 1 : 0;
        p.tail_merged = 1;
        p.tail_solve = p.solve - 1;
        p.ustar_cur = h->d_ustar2[cur3]; p.stats_cur = h->d_stats2[cur3];
        p.ustar_prev = h->d_ustar2[prev3]; p.stats_prev = h->d_stats2[prev3];
        p.ticket = h->d_ticket;
        hipStream_t st = h->stream;
        p.flag_tail = h->d_flags + kSlots * B * bn::kFlagStride;
        if (p.have_prev) { p.wait_tail = h->tails; h->tails += 1; }
        if (h->ticket_overlap && overlap) {
            // member of an overlapped batch (as in the pipelined branch above): the merge of this launch counts itself into its slot
            // (one count per solve), the next launch -- rollouts and tail -- waits for that count instead of for this kernel's end
            p.flag_part = h->d_flags;
            p.err = h->d_err; p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 2) * bn::kFlagStride);
            if (h->arm_snap) { p.mean_snap = h->d_mean_snap; h->arm_snap = false; }
            p.cur_slot = cur3; p.prev_slot = prev3;
            p.wait_part = h->pub[prev3];
            if (alt_buffers) {
                if (h->d_Xalt[alt_buffers - 1]) p.X = h->d_Xalt[alt_buffers - 1];
                if (h->d_Ualt[alt_buffers - 1] && p.U) p.U = h->d_Ualt[alt_buffers - 1];
            }
            p.overlap = (h->prev_published && p.have_prev) ? 1 : 0;
            if (on_stream) st = on_stream;
            if (p.overlap) h->overlap_used = true;
            h->pub[cur3] += 1;                                         // one count per instance and solve
            h->prev_published = true;
        }
        if (p.slip_on) BN_HIP(bn::launch_rollout_sampled(p, mode, st));
        else BN_HIP(bn::launch_rollout(p, mode, st));
        if (prof_grouped) h->prof_in_group = (h->prof_in_group + 1) % kProfGroup;
        h->solves += 1;
        h->tail_pending = true;
        return BN_OK;
    }
    // library-enqueued exchange of a K-sharded solve: the shard's rows go straight to their place among the gathered ones (in-place all-gather)
    if (shard_rollout && h->shard_inplace) p.part = h->d_gathered + (size_t)h->shard_rank * p.nblk * (2 + 2 * T);
    if (p.slip_on) {
        BN_HIP(bn::launch_rollout_sampled(p, mode, h->stream));
    } else {
        if (h->slow_path) p.wave_kernel = 1;                 // no control tile in LDS, any horizon; REF arithmetic when p.ref_order
        else if (!p.store_u) p.U = nullptr;                  // (the buffer may exist for the throughput kernel only)
        BN_HIP(bn::launch_rollout(p, mode, h->stream));
    }
    if (prof) BN_HIP(hipEventRecord(ev[1], h->stream));
    if (shard_rollout) {
        if (prof) BN_HIP(hipEventRecord(ev[2], h->stream));
        h->solves += 1;
        h->tail_pending = false;
        h->shard_pending = true;
        return BN_OK;
    }
    p.state = p.state_copy;
    BN_HIP(bn::launch_finish(p, h->stream));
    if (prof) BN_HIP(hipEventRecord(ev[2], h->stream));
    h->solves += 1;
    h->tail_pending = false;
    return BN_OK;
}

int bn_mppi_solve_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps,
                        bn_noise_kind noise)
{
    if (h && !h->replaying && !h->journal.empty()) {
        // a single solve behind overlapped batches whose error word has not been checked yet: journalled as a batch of one
        h->last_batch_overlapped = false;
        journal_arm_states(h);
        journal_push(h, bn_mppi::BatchRec{1, states, eps, noise, 1, 0, false, nullptr}, states_where == BN_MEM_DEVICE && noise != BN_NOISE_HOST_KT2);
    }
    return solve_impl(h, states, states_where, eps, noise, false);
}

#ifdef BN_TIMING
// host-side split of the host-paced forward (tools/stamps_forward.py prints it at exit): nanoseconds and calls per piece
struct HpClock { const char *name; long long ns = 0, n = 0; };
static HpClock g_hpc[6] = {{"post: request words"}, {"post: hipStreamWaitEvent"}, {"prelaunch: checks + ack"}, {"prelaunch: kernel launch"}, {"prelaunch: hipEventRecord"}, {"self_check + bind"}};
struct HpTick { int i; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                explicit HpTick(int i_) : i(i_) {}
                ~HpTick() { g_hpc[i].ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_hpc[i].n += 1; } };
#define BN_HP_TICK(i) HpTick bn_hp_tick_##i(i)
extern "C" void bn_mppi_debug_hp_clock(void)
{
    for (const HpClock &c : g_hpc) if (c.n) std::fprintf(stderr, "[hp host] %-28s %7.2f us x %lld\n", c.name, c.ns / 1e3 / c.n, c.n);
}
#else
#define BN_HP_TICK(i) do { } while (0)
#endif

// Host-paced loop: enqueue the launch of the solve AFTER the latest one, on a private stream; it waits on the device for its state.
// Best effort: returns false (and leaves nothing behind) when the handle's state does not allow it.
static bool hp_prelaunch(bn_mppi *h)
{
    if (!h->hp_enabled || h->hp_armed || h->self_off || !h->hp_gran_valid || h->tail_pending || h->shard_pending || h->in_episode || h->replaying ||
        !h->map_set || !h->goal_set || h->solves == 0)
        return false;
    std::chrono::steady_clock::time_point bn_t_pre = std::chrono::steady_clock::now(); (void)bn_t_pre;
    {   // not beside another handle's launches (the latency kernel is sized for a device it has to itself: see g_handles)
        std::lock_guard<std::mutex> lock(g_overlap_mu);
        for (bn_mppi *o : g_handles[h->cfg.device_id]) {
            if (o == h) continue;
            for (int q = 0; q < kMaxStreams; ++q) {
                const hipStream_t os = q ? o->xstream[q - 1] : o->stream;
                if ((q && !os) || os == h->stream) continue;
                if (hipStreamQuery(os) != hipSuccess) { (void)hipGetLastError(); return false; }
            }
        }
        (void)hipGetLastError();
    }
    bn::SolveParams p = h->p;
    const size_t B = 1;
    const int cur3 = (int)(h->solves % kSlots), prev3 = (int)((h->solves + kSlots - 1) % kSlots);
    p.solve = h->solves; p.tail_solve = p.solve;
    p.part = h->d_part[cur3]; p.cost = h->d_cost[cur3]; p.state_copy = h->d_state_copy[cur3];
    p.part_prev = h->d_part[prev3]; p.cost_prev = h->d_cost[prev3]; p.state_prev = h->d_state_copy[prev3];
    p.have_prev = 0; p.mean_from_part = 1;             // the mean IS the merge of the previous solve's rows (its tail writes the same values)
    p.lat_kernel = 1; p.wave_kernel = 0;
    p.flag_tail = h->d_flags + kSlots * B * bn::kFlagStride;
    p.flag_part = h->d_flags;
    p.err = h->d_err; p.err_dev = reinterpret_cast<int *>(h->d_flags + ((kSlots + 1) * B + 2) * bn::kFlagStride);
    p.cur_slot = cur3; p.prev_slot = prev3;
    p.wait_part = h->pub[prev3];
    p.gran = h->d_gran[cur3]; p.gran_prev = h->d_gran[prev3];
    p.overlap = 1;
    p.self_tail = 1;
    p.wait_part_self = h->pub[cur3] + (unsigned long long)p.nblk;
    p.wait_tail_self = h->tails;
    const int xi = 1 - h->x_idx;                       // the other trajectory / control buffer: two launches in flight never write the same addresses
    if (xi && h->d_Xalt[0]) p.X = h->d_Xalt[0];
    if (!p.store_u) p.U = nullptr;
    else if (xi && h->d_Ualt[0]) p.U = h->d_Ualt[0];
    p.host_paced = 1;
    if (exp_env("BN_NO_EARLY_MAIL")) p.no_early_mail = 1;      // (experiment builds)
    p.sv0 = h->hp_prev_state[0]; p.sv1 = h->hp_prev_state[1]; p.sv2 = h->hp_prev_state[2];
    p.spec_extra = h->hp_extra;
    const uint32_t tag = h->hp_seq + 1;
    const int slot = (int)(tag % kSlots);
    static_assert(kSlots == 4, "the kernels acknowledge a request in word 1 + (tag & 3)");
    if (tag > (uint32_t)kSlots) {
        // The slot's previous user -- the launch four tags back -- must have taken (or given up on) its request before the words are
        // written again: its tail workgroup says so.  In a loop paced by bn_mppi_first_action that was three solves ago; a caller that
        // fires forwards back to back is held to the device's pace here (bounded: a launch gives up by itself after ~50 ms).
        const unsigned long long *ack = h->h_req + (size_t)kSlots * 8 + 1 + slot;
        const auto t0 = std::chrono::steady_clock::now();
        for (long it = 0; __atomic_load_n(ack, __ATOMIC_ACQUIRE) != (unsigned long long)(tag - kSlots); ++it)
            if ((it & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) { h->hp_enabled = false; return false; }
    }
    h->hp_seq = tag;
    p.req_tag = tag;
    p.req_direct = h->req_bar ? 1 : 0;
    // the launch's patience, in looks: ~2 us each over PCIe, ~0.7 us in device memory -- at least 50 ms either way; the host stays
    // below 20 ms (hp_post looks at the clock), so a "go" never meets a "gave up"
    p.req_polls = (h->req_bar && h->hp_polls == 25000) ? 100000 : h->hp_polls;
    p.req_host = (h->req_bar ? h->req_bar : h->d_req) + (size_t)slot * 8;
    p.req_dev = h->d_req_dev + (size_t)slot * 8;
    p.spec_status = h->d_req + (size_t)kSlots * 8;
    p.state = h->d_state;                              // (not read: the state comes with the request)
    const int q = h->hp_next_q;
#ifdef BN_TIMING
    g_hpc[2].ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - bn_t_pre).count(); g_hpc[2].n += 1;
#endif
    hipError_t e;
    // (the completion event goes out with the launch: one runtime call)
    { BN_HP_TICK(3); e = p.ref_order ? bn::launch_rollout_lat_host_ref(p, h->hp_stream[q], h->hp_ev[q]) : bn::launch_rollout_lat_host(p, h->hp_stream[q], h->hp_ev[q]); }
    if (e != hipSuccess) { (void)hipGetLastError(); h->hp_enabled = false; return false; }
    h->pub[cur3] += (unsigned long long)p.nblk;
    h->tails += 1;
    h->solves += 1;
    h->hp_armed = true; h->hp_tag = tag; h->hp_slot = slot; h->hp_q = q; h->hp_xidx = xi;
    h->hp_armed_at = std::chrono::steady_clock::now();
    h->hp_next_q = 1 - q;
    return true;
}

// ... and hand the waiting launch its state.
static int hp_post(bn_mppi *h, const float st[3], float *out_device)
{
    // The waiting launch runs on a stream of its own: it is ordered behind the HOST (this store), not behind what the caller enqueued on
    // the handle's stream before this call.  The contract of the mode (header): work left there that still reads the planner's own
    // buffers -- weights, trajectory batch, the shared U* | X* block -- or the last users of the memory `out_device` was carved from must
    // have completed; fresh output blocks per step (what the drop-in class hands out) need nothing.  (Asking the stream -- hipStreamQuery
    // -- is no way out: the event that orders the previous solve's launch in front of the caller's consumers keeps it "busy" for tens of
    // microseconds after that launch has ended, and every step would wait for it: 17 -> 39 us, measured.)
    { BN_HP_TICK(0); hp_write_request(h, h->hp_slot, h->hp_tag, st, out_device, 1u); }
    h->hp_armed = false;
    h->hp_posted_tag = h->hp_tag;
    std::memcpy(h->hp_posted_state, st, 12);
    h->hp_posted_out = out_device;
    std::memcpy(h->hp_prev_state, st, 12);
    h->x_idx = h->hp_xidx;
    h->map_epoch_at_solve = h->map_epoch;
    h->last_eps = nullptr; h->last_mode = bn::kEpsPhilox;
    h->tail_pending = false; h->prev_published = false; h->last_batch_overlapped = false;
    h->self_used = true;
    h->hp_gran_valid = true;
    return BN_OK;
}

// Whatever the caller enqueues on the handle's stream from here on is ordered behind the posted solve's launch (its outputs).  Issued
// BEHIND the next solve's prelaunch: that launch needs every microsecond of head start it can get (it has to be through its prologue when
// the host comes back with the next state), this call only has to be made before the caller gets control back.
static int hp_order_outputs(bn_mppi *h, int q)
{
    if (h->cfg.flags & BN_FLAG_UNORDERED_OUTPUTS) { h->hp_unordered = true; h->hp_unordered_q = q; return BN_OK; }    // made up for by the first call that needs it
    if (!exp_env("BN_HP_NO_WAIT")) { BN_HP_TICK(1); BN_HIP(hipStreamWaitEvent(h->stream, h->hp_ev[q], 0)); }      // (the switch: experiment builds, tools/stamps_forward.py)
    return BN_OK;
}

// MPPI.forward as ONE call -- and, on the latency kernel, ONE launch: the solve's tail (merge -> U* -> first-action mailbox -> X* ->
// weights) rides in the rollout launch as a second aux workgroup that waits on the device for the rollout workgroups of its own launch.
static int forward_impl(bn_mppi_t *h, const float *states_device, const float *state_host, const float *eps_device, bn_noise_kind noise,
                        float *out_device)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!states_device && !state_host) return fail(BN_ERR_INVALID, "states is null");
    if ((noise == BN_NOISE_PHILOX) != (eps_device == nullptr)) return fail(BN_ERR_INVALID, "eps must be NULL exactly when noise == BN_NOISE_PHILOX");
    const bool paced = h->hp_enabled && state_host && noise == BN_NOISE_PHILOX && !h->self_off;
    // (the host took longer than the launch waits -- or, with the request words polled by every workgroup of the launch, may be about
    // to: a "go" must never meet a "gave up" -- : start over below)
    if (paced && h->hp_armed && !h->hp_skip_check &&
        (hp_gave_up(h, h->hp_tag) || std::chrono::steady_clock::now() - h->hp_armed_at > std::chrono::milliseconds(20))) hp_cancel(h);
    if (paced && h->hp_armed) {                        // the loop's steady state: the launch is there and waits for exactly this
        BN_BIND_Q(h);
        if (int rc = self_check(h)) return rc;
        const int q_posted = h->hp_q;
        if (int rc = hp_post(h, state_host, out_device)) return rc;
        (void)hp_prelaunch(h);                         // ... and the next one goes out while this one runs
        return hp_order_outputs(h, q_posted);
    }
    {
        BN_BIND(h);
        if (int rc = settle_point(h)) return rc;       // behind overlapped batches: those are checked (one synchronisation) first
        // a tail still pending (an earlier bn_mppi_solve_async): written by its own kernel first -- the one-launch path carries no other
        // tail than its own (two tails of one launch would write the outputs from two workgroups with different store scopes)
        if (h->lat_kernel && h->tail_pending && !h->self_off) { if (int rc = flush_tail(h)) return rc; }
    }
    h->self_out_copy = out_device;
    // by value: one instance on the latency kernel travels in the kernel arguments; anything else is staged and uploaded (bn_mppi_solve_async's host path)
    const bool by_value = state_host && h->p.B == 1 && h->lat_kernel && !h->self_off && !h->tail_pending && !(h->cfg.flags & BN_FLAG_NO_PIPELINE);
    h->inline_state = by_value ? state_host : nullptr;
    const float *st = by_value ? h->d_state : (state_host ? state_host : states_device);
    const int rc0 = solve_impl(h, st, (state_host && !by_value) ? BN_MEM_HOST : BN_MEM_DEVICE, eps_device, noise, false, false, nullptr, 0, true);
    h->self_out_copy = nullptr;
    h->inline_state = nullptr;
    if (rc0) return rc0;
    if (h->self_tail_launched) {                       // tail and the caller's copy ride in the launch
        if (paced && by_value) {
            BN_BIND(h);
            std::memcpy(h->hp_prev_state, state_host, 12);
            (void)hp_prelaunch(h);
        }
        return BN_OK;
    }
    BN_BIND(h);
    if (!h->tail_pending && out_device) {              // two-launch modes: the tail has run; one small copy on the stream
        const size_t n = (size_t)h->p.B * ((size_t)h->p.T * 2 + ((size_t)h->p.T + 1) * 3);
        BN_HIP(hipMemcpyAsync(out_device, h->d_ustar, n * 4, hipMemcpyDeviceToDevice, h->stream));
        return BN_OK;
    }
    return flush_tail(h, out_device);
}

// The launch a request was posted to had given up a moment before (its tail workgroup said so after the post): the solve never ran.
// Take it -- and a successor prelaunched behind it -- out of the books and run it again the ordinary way, from the same state.
static int hp_redo(bn_mppi *h)
{
    hp_cancel(h);
    h->solves -= 1;
    h->pub[h->solves % kSlots] -= (unsigned long long)h->p.nblk;
    h->tails -= 1;
    h->hp_posted_tag = 0;
    h->hp_gran_valid = false;
    float st[3];
    std::memcpy(st, h->hp_posted_state, 12);
    return forward_impl(h, nullptr, st, nullptr, BN_NOISE_PHILOX, h->hp_posted_out);
}

int bn_mppi_forward_async(bn_mppi_t *h, const float *states_device, const float *eps_device, bn_noise_kind noise, float *out_device)
{
    // MPPI.forward as ONE call for a host loop that consumes every solve's outputs (test_mppi.py:174-183): the solve and
    // its tail, both only enqueued; U*, X*, weights are in the device buffers in stream order.
    if (!states_device) return fail(BN_ERR_INVALID, "states is null");
    return forward_impl(h, states_device, nullptr, eps_device, noise, out_device);
}

int bn_mppi_forward_state_async(bn_mppi_t *h, const float *states_host, const float *eps_device, bn_noise_kind noise, float *out_device)
{
    if (!states_host) return fail(BN_ERR_INVALID, "states is null");
    return forward_impl(h, nullptr, states_host, eps_device, noise, out_device);
}

int bn_mppi_set_rollout_offset(bn_mppi_t *h, int64_t first_rollout)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (first_rollout < 0 || first_rollout + h->p.K > 0x7fffffffLL) return fail(BN_ERR_INVALID, "first_rollout out of range");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    h->p.k0 = (int)first_rollout;
    return BN_OK;
}

int bn_mppi_shard_rollout_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps,
                                bn_noise_kind noise)
{
    if (h && h->p.B != 1) return fail(BN_ERR_INVALID, "a K-sharded solve takes one instance per handle");
    if (h && h->p.slip_on) return fail(BN_ERR_INVALID, "K-sharding is not available in sampled-slip mode");
    return solve_impl(h, states, states_where, eps, noise, true);
}

int bn_mppi_shard_partials(bn_mppi_t *h, const float **partials_device, int32_t *workgroups, int32_t *floats_per_workgroup)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->shard_pending) return fail(BN_ERR_STATE, "no sharded solve in flight");
    const int cur = (int)((h->solves - 1) % kSlots);
    if (partials_device) *partials_device = h->d_part[cur];
    if (workgroups) *workgroups = h->p.nblk;
    if (floats_per_workgroup) *floats_per_workgroup = 2 + 2 * h->p.T;
    return BN_OK;
}

int bn_mppi_shard_finish_async(bn_mppi_t *h, const float *all_partials_device, int32_t total_workgroups)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->shard_pending) return fail(BN_ERR_STATE, "no sharded solve in flight");
    if (!all_partials_device || total_workgroups < h->p.nblk) return fail(BN_ERR_INVALID, "need the partials of every shard");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    bn::SolveParams p = h->p;
    const int cur = (int)((h->solves - 1) % kSlots);
    p.solve = p.tail_solve = h->solves - 1;
    p.part = const_cast<float *>(all_partials_device);      // merged in shard order: identical on every rank
    p.nblk = total_workgroups;
    p.cost = h->d_cost[cur]; p.state = h->d_state_copy[cur];
    BN_HIP(bn::launch_finish(p, h->stream));                 // weights of the local rollouts, normalised by the global sum
    h->shard_pending = false;
    return BN_OK;
}

// ---- the exchange of a K-sharded solve enqueued by the library: RCCL on the handle's stream -----------------------------------
// Round 4's ShardedMPPI drove the three steps from Python -- launch, torch.distributed.all_gather_into_tensor (its own stream, two
// events to fence it against the planner's), launch: 60 us per solve on one rank against 28 unsharded.  Here the triple is ONE C call
// and ONE queue: rollout kernel, ncclAllGather, tail kernel, stream-ordered, no host wait, no event.  RCCL is opened with dlopen on
// first use: the library has no link-time dependency on it, and a process that never shards a solve never loads it.
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
const RcclApi *rccl_api()
{
    static const RcclApi api = [] {
        RcclApi a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (!a.lib) return a;
        a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(a.lib, "ncclAllGather"));
        a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.get_unique_id || !a.comm_init_rank || !a.all_gather || !a.comm_destroy || !a.error_string) a.lib = nullptr;
        return a;
    }();
    return api.lib ? &api : nullptr;
}
}  // namespace

static void shard_comm_release(bn_mppi *h)
{
    if (h->shard_side) (void)hipStreamSynchronize(h->shard_side);
    if (h->shard_comm) { if (const RcclApi *r = rccl_api()) (void)r->comm_destroy(static_cast<ncclComm_t>(h->shard_comm)); }
    h->shard_comm = nullptr;
    h->shard_prepared = false;
    if (h->d_gathered) (void)hipFree(h->d_gathered);
    if (h->d_shard_merged) (void)hipFree(h->d_shard_merged);
    h->d_gathered = h->d_shard_merged = nullptr;
    if (h->ev_shard_merge) (void)hipEventDestroy(h->ev_shard_merge);
    for (hipEvent_t &e : h->ev_shard_tail) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    h->ev_shard_merge = nullptr;
    if (h->shard_side_own && h->shard_side) (void)hipStreamDestroy(h->shard_side);
    h->shard_side = nullptr;
}

int bn_dist_unique_id(uint8_t out[BN_DIST_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == BN_DIST_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!out) return fail(BN_ERR_INVALID, "null output");
    const RcclApi *r = rccl_api();
    if (!r) return fail(BN_ERR_HIP, "librccl.so could not be opened: %s", dlerror());
    ncclUniqueId id;
    const ncclResult_t e = r->get_unique_id(&id);
    if (e != ncclSuccess) return fail(BN_ERR_HIP, "ncclGetUniqueId: %s", r->error_string(e));
    std::memcpy(out, &id, sizeof(id));
    return BN_OK;
}

int bn_mppi_shard_comm_prepare(bn_mppi_t *h, int32_t world_size, int32_t rank)
{
    // Everything bn_mppi_shard_comm_init can fail on BEFORE it enters the collective (ncclCommInitRank): opening RCCL, the handle's
    // state, the buffers, events and the side stream.  Local, no communication: the ranks of a job call it, agree on the outcome by
    // whatever means they have (benchnav_amd.sharding: one all-reduce over the torch group), and enter the collective only if every
    // rank is ready -- a rank that cannot must not leave the others waiting in ncclCommInitRank's bootstrap.
    if (!h) return fail(BN_ERR_INVALID, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(BN_ERR_INVALID, "rank %d outside a world of %d", rank, world_size);
    if (h->p.B != 1 || h->p.slip_on) return fail(BN_ERR_INVALID, "a K-sharded solve takes one instance per handle, without sampled slip");
    if (h->shard_comm) return fail(BN_ERR_STATE, "the handle has a communicator already");
    if ((size_t)world_size * h->p.nblk > 1024) return fail(BN_ERR_INVALID, "the library's exchange merges at most 1024 workgroups (65536 rollouts)");
    if (h->shard_prepared) {
        if (h->shard_world == world_size && h->shard_rank == rank) return BN_OK;
        return fail(BN_ERR_STATE, "the handle was prepared as rank %d of %d", h->shard_rank, h->shard_world);
    }
    const RcclApi *r = rccl_api();
    if (!r) return fail(BN_ERR_HIP, "librccl.so could not be opened: %s", dlerror());
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t bytes = (size_t)world_size * h->p.nblk * (2 + 2 * (size_t)h->p.T) * sizeof(float);
    if (hipMalloc((void **)&h->d_gathered, bytes) != hipSuccess) {
        (void)hipGetLastError();
        h->d_gathered = nullptr;
        return fail(BN_ERR_HIP, "hipMalloc of %zu B for the gathered partials failed", bytes);
    }
    h->shard_world = world_size;
    h->shard_rank = rank;
    // the side stream of the tail: the handle's second stream if it has one (overlapped batches), else one of its own
    // [ kSlots merged rows | 64 group rows | ticket ]
    const size_t merged_floats = kSlots * (2 * (size_t)h->p.T + 2) + 64 * (2 + 2 * (size_t)h->p.T) + 4;
    bool ok = hipMalloc((void **)&h->d_shard_merged, merged_floats * sizeof(float)) == hipSuccess &&
              hipMemset(h->d_shard_merged, 0, merged_floats * sizeof(float)) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_shard_merge, hipEventDisableTiming) == hipSuccess;
    for (int q = 0; ok && q < kSlots; ++q) ok = hipEventCreateWithFlags(&h->ev_shard_tail[q], hipEventDisableTiming) == hipSuccess;
    if (ok && h->n_streams > 1 && h->xstream[0]) h->shard_side = h->xstream[0];
    else if (ok) { ok = hipStreamCreateWithFlags(&h->shard_side, hipStreamNonBlocking) == hipSuccess; h->shard_side_own = ok; }
    if (!ok) { (void)hipGetLastError(); shard_comm_release(h); return fail(BN_ERR_HIP, "stream / event / buffer creation for the sharded solve failed"); }
    h->shard_prepared = true;
    return BN_OK;
}

int bn_mppi_shard_comm_init(bn_mppi_t *h, const uint8_t unique_id[BN_DIST_UNIQUE_ID_BYTES], int32_t world_size, int32_t rank)
{
    if (!h || !unique_id) return fail(BN_ERR_INVALID, "null argument");
    if (int rc = bn_mppi_shard_comm_prepare(h, world_size, rank)) return rc;      // (a no-op behind the caller's own prepare)
    const RcclApi *r = rccl_api();
    BN_BIND(h);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->comm_init_rank(&comm, world_size, id, rank);      // collective: every rank of the solve calls it
    if (e != ncclSuccess) { shard_comm_release(h); return fail(BN_ERR_HIP, "ncclCommInitRank: %s", r->error_string(e)); }
    h->shard_comm = comm;
    return BN_OK;
}

int bn_mppi_shard_solve_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps, bn_noise_kind noise)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->shard_comm) return fail(BN_ERR_STATE, "bn_mppi_shard_comm_init must precede bn_mppi_shard_solve_async");
    // The rollouts of THIS solve do not wait for the previous solve's tail on the side stream (flush_tail would join it): they read
    // the mean the merge wrote, on this stream.  And they write their partial rows where the all-gather wants them (in place).
    // Every per-solve buffer rotates over kSlots solves -- the costs and the start state the rollouts below write, the merged row -- and
    // the side-stream tail of the solve kSlots back still reads its slot (weights, cost copy, X*).  That tail is done unless something is
    // badly stuck, so the host LOOKS (hipEventQuery) and only a tail still running puts a cross-queue wait in front of the ROLLOUTS
    // (the check used to sit behind them and covered the merged row only) -- an event wait between queues costs
    // microseconds on the handle's stream even when the event has long fired (first cut of this path: 72 us per solve with an
    // unconditional wait against 47 without the split).
    {
        BN_BIND(h);
        const int next = (int)(h->solves % kSlots);
        if (h->solves >= (uint64_t)kSlots && h->ev_shard_tail[next] && hipEventQuery(h->ev_shard_tail[next]) != hipSuccess) {
            (void)hipGetLastError();
            BN_HIP(hipStreamWaitEvent(h->stream, h->ev_shard_tail[next], 0));
        }
    }
    const bool inflight = h->shard_tail_inflight;
    h->shard_tail_inflight = false;
    h->shard_inplace = true;
    const int rc0 = bn_mppi_shard_rollout_async(h, states, states_where, eps, noise);
    h->shard_inplace = false;
    h->shard_tail_inflight = inflight;
    if (rc0) return rc0;
    BN_BIND(h);
    const int cur = (int)((h->solves - 1) % kSlots);
    const size_t count = (size_t)h->p.nblk * (2 + 2 * (size_t)h->p.T);          // equal shards: every rank contributes the same rows
    const RcclApi *r = rccl_api();
    const ncclResult_t e = r->all_gather(h->d_gathered + (size_t)h->shard_rank * count, h->d_gathered, count, ncclFloat,
                                         static_cast<ncclComm_t>(h->shard_comm), h->stream);
    if (e != ncclSuccess) {
        h->shard_pending = false;                            // the solve is lost, the handle is not: the next call starts a new one
        return fail(BN_ERR_HIP, "ncclAllGather: %s", r->error_string(e));
    }
    // Merge on the handle's stream -- U*, the next mean, the softmin statistics: all the next solve's rollouts wait for -- and the
    // rest of the tail (X* rollout, the shard's weights, the cost copy: ~10 us of one workgroup) on the side stream, beside them.
    // What the two streams share: the outputs (tails stay in order on their one stream; U* is the merge's alone), the per-solve slots
    // of costs / states, and the merged row (written by the merge, read by the tail) -- all rotating over kSlots solves and covered by
    // the look at the tail kSlots back in front of the rollouts, above.
    if (int rc = settle_point(h)) return rc;
    bn::SolveParams p = h->p;
    p.solve = p.tail_solve = h->solves - 1;
    p.part = h->d_gathered;                                  // merged in shard order: identical on every rank
    p.nblk = h->shard_world * h->p.nblk;
    p.cost = h->d_cost[cur]; p.state = h->d_state_copy[cur];
    float *merged = h->d_shard_merged + (size_t)cur * (2 * (size_t)p.T + 2);
    p.ustar_cur = merged; p.stats_cur = merged + 2 * (size_t)p.T;
    float *group_rows = h->d_shard_merged + kSlots * (2 * (size_t)p.T + 2);
    BN_HIP(bn::launch_shard_merge(p, group_rows, reinterpret_cast<int *>(group_rows + 64 * (2 + 2 * (size_t)p.T)), h->stream));
    BN_HIP(hipEventRecord(h->ev_shard_merge, h->stream));
    BN_HIP(hipStreamWaitEvent(h->shard_side, h->ev_shard_merge, 0));
    p.tail_merged = 1;
    p.ustar_written = 1;                                     // (the merge kernel above wrote U*: see SolveParams::ustar_written)
    p.ustar_prev = merged; p.stats_prev = merged + 2 * (size_t)p.T;
    BN_HIP(bn::launch_finish(p, h->shard_side));            // weights of the local rollouts, normalised by the global sum; X*
    BN_HIP(hipEventRecord(h->ev_shard_tail[cur], h->shard_side));
    h->shard_tail_slot = cur;
    h->shard_tail_inflight = true;
    h->shard_pending = false;
    return BN_OK;
}

int bn_mppi_solve_n_async(bn_mppi_t *h, int32_t n, const float *states, bn_mem_kind states_where, const float *eps,
                          bn_noise_kind noise, int32_t eps_ring, int64_t eps_stride)
{
    if (n < 0) return fail(BN_ERR_INVALID, "n must be >= 0");
    if (eps && (eps_ring < 1 || eps_stride < 0)) return fail(BN_ERR_INVALID, "eps_ring must be >= 1 and eps_stride >= 0");
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    // Overlapped launches.  Solve i+1 needs only the softmin partials of solve i, yet in one stream it also waits for solve i's
    // kernel to drain and for the dispatch of its own (~2.4 us of a 13 us step).  With the latency kernel consecutive solves
    // alternate between the handle's stream and a second one: solve i+1 is dispatched while solve i runs, stages its window and
    // draws its first noise, and then waits on a device counter for solve i's partials (published as device-scope stores by
    // the workgroups themselves).  At most two launches are in flight (each stream serialises its own); per-solve buffers
    // rotate over three slots so that what a launch overwrites was last read by a launch that has completed on its stream;
    // the tails (aux workgroups) are ordered by a second counter.  Results are bit-identical to the one-stream chain.
    // Device-side episodes overlap the same way (role kernel): the successor reads the state its predecessor advanced with
    // device-scope loads after its wait (64 instances: 29.6 -> 24.8 us per control step).
    const bool overlap = (h->lat_kernel || h->role_overlap || h->ticket_overlap) && h->n_streams > 1 && (h->d_Xalt[0] || h->p.lean) && n >= 3 && states_where == BN_MEM_DEVICE &&
                         noise != BN_NOISE_HOST_KT2 && !(h->cfg.flags & (BN_FLAG_PROFILE | BN_FLAG_NO_OVERLAP)) && !h->shard_pending && !h->overlap_off;
    // which mode this batch runs in: the handle's choice, or -- for its first 192 launches, once -- the other one, for a look (tune_record)
    int mode = overlap ? h->tune.chosen : 1;
    if (overlap && h->tune.explore && n >= 128 && !h->in_episode && !h->replaying) {
        const int32_t ring = eps ? eps_ring : 1;
        const int32_t prefix = ((192 + ring - 1) / ring) * ring;       // (a whole number of turns of the caller's noise ring: the rest starts at block 0 again)
        if (n > prefix + 2) {
            if (int rc = bn_mppi_solve_n_async(h, prefix, states, states_where, eps, noise, eps_ring, eps_stride)) return rc;
            h->tune.explore = 0;                                        // (looked, whether or not the windows were conclusive)
            return bn_mppi_solve_n_async(h, n - prefix, states, states_where, eps, noise, eps_ring, eps_stride);
        }
        mode = 1 - h->tune.chosen;
    }
    if (mode != h->run_mode) h->ov_sample.valid = false;
    h->run_mode = mode;
    bool mine = overlap && mode == 0;
    if (mine) {                                         // see g_overlap_owner
        BN_BIND(h);
        std::lock_guard<std::mutex> lock(g_overlap_mu);
        bn_mppi *&owner = g_overlap_owner[h->cfg.device_id];
        if (owner && owner != h) {                      // is the owner's latest overlapped batch still in flight?
            bool busy = hipStreamQuery(owner->stream) != hipSuccess;
            for (int q = 0; !busy && q + 1 < owner->n_streams; ++q) busy = hipStreamQuery(owner->xstream[q]) != hipSuccess;
            (void)hipGetLastError();
            if (busy) mine = false;
        }
        for (bn_mppi *o : g_handles[h->cfg.device_id]) {   // ... and is every other handle idle?  (see g_handles)
            if (!mine) break;
            if (o == h) continue;
            for (int q = 0; mine && q < kMaxStreams; ++q) {            // (incl. the second private stream of a host-paced handle)
                const hipStream_t os = q ? o->xstream[q - 1] : o->stream;
                if ((q && !os) || os == h->stream) continue;
                if (hipStreamQuery(os) != hipSuccess) mine = false;
            }
        }
        (void)hipGetLastError();
        const size_t slots_ = h->wave_kernel ? 24 * (size_t)std::max(h->n_cus, 1) : (h->lat_kernel ? (size_t)std::max(h->n_cus, 1) : h->resident_wgs);
        if (mine && 2 * (size_t)h->p.B * (h->p.nblk + 1) > slots_ && !own_device_for_big_overlap(h->cfg.device_id)) mine = false;
        // A launch that needs well over one residency round by itself is never fully placed while it runs: its successor's waiting
        // workgroups compete with its OWN remaining ones for every slot that frees up, whatever the host orders.  Up to a quarter over
        // (64 instances of K = 1024: 1088 workgroups on 1024 slots; 70: 1190) the remainder is placed when the first round drains and
        // nothing has ever expired (tens of thousands of launches, fresh processes included); at one and a half rounds (3 instances of
        // K = 8192 at one workgroup per CU: 387 on 256) the first batch of a fresh process expired every time (round 4,
        // tests/differential.py `ops` seed 2681).  Such batches run on one stream.
        if (mine && 4 * (size_t)h->p.B * (h->p.nblk + 1) > 5 * slots_) mine = false;
        if (mine) owner = h;
        else if (!owner) g_overlap_owner.erase(h->cfg.device_id);      // (operator[] above created an empty entry)
        g_overlap_owners.store((int)g_overlap_owner.size(), std::memory_order_relaxed);
    }
    const bn_mppi::BatchRec rec{n, states, eps, noise, eps_ring, eps_stride, false, nullptr};
    const bool replayable = states_where == BN_MEM_DEVICE && noise != BN_NOISE_HOST_KT2;
    if (replayable && !h->in_episode) journal_arm_states(h);
    if (!mine) {
        h->last_batch_overlapped = false;
        for (int32_t i = 0; i < n; ++i) {
            const float *e = eps ? eps + (size_t)(i % eps_ring) * (size_t)eps_stride : nullptr;
            if (int rc = solve_impl(h, states, states_where, e, noise, false)) return rc;
            if (overlap && (i & 63) == 63) tune_sample(h);
        }
        if (!h->in_episode) journal_push(h, rec, replayable);   // behind an overlapped batch not yet checked: lost with it, re-run with it
        return BN_OK;
    }
    BN_BIND(h);
    h->last_batch_overlapped = true;
    if (!h->replaying && h->journal.empty() && !h->journal_lost) {   // first batch since the error word was last seen clean
        h->journal_solves0 = h->solves;
        h->arm_snap = true;
    }
    const int S = h->n_streams;
    // The scheme relies on a launch having been dispatched before its successor becomes eligible: waiting workgroups hold their
    // slots, and the hardware does not share freed slots fairly between two queues (seen with two 70-instance launches made
    // eligible at the same instant: the successor's waiting workgroups took every slot, the predecessor's never got one, the
    // waits expired).  Inside a batch that holds by construction -- launch i+1 becomes eligible when launch i-1 completes, a whole
    // kernel after launch i did.  At the START it holds when the stream is idle (solve 1 is enqueued microseconds after solve 0
    // went out); if earlier work is still pending, solves 0 and 1 would become eligible together, so launches big enough to
    // crowd each other out wait for the stream first.
    const size_t slots = h->wave_kernel ? 24 * (size_t)std::max(h->n_cus, 1) : (h->lat_kernel ? (size_t)std::max(h->n_cus, 1) : h->resident_wgs);
    bool idle = hipStreamQuery(h->stream) == hipSuccess;
    (void)hipGetLastError();
    if (!idle && 2 * (size_t)h->p.B * (h->p.nblk + 1) > slots) {
        BN_HIP(hipStreamSynchronize(h->stream));
        idle = true;
    }
    if (!idle) {                                                    // fork: the extra streams start behind everything enqueued so far
        BN_HIP(hipEventRecord(h->ev_fork, h->stream));              // (nothing is: no event packet in front of the first launches)
        for (int q = 0; q + 1 < S; ++q) BN_HIP(hipStreamWaitEvent(h->xstream[q], h->ev_fork, 0));
    }
    // The last launch of a long batch carries its own tail as a second aux workgroup (SolveParams::self_tail; the one-launch kernel,
    // rollout_lat.inc mode 1) instead of a tail kernel behind it: since round 6 that workgroup stages its window while the rollouts run
    // and polls the granules of their rows -- 3 us off a 20-solve region (tools/region_ab.py: 215.2 -> 212.2 us; the counter-waiting
    // tail of rounds 2-5 had measured 5 us WORSE than the kernel behind).  BN_NO_SELF_TAIL (experiment builds) turns it off.
    const bool exp_self_tail = exp_env("BN_NO_SELF_TAIL") == nullptr;      // (read per batch: tools/region_ab.py toggles it inside one process)
    static const bool exp_align = exp_env("BN_NO_ALIGN") == nullptr;
    if (!exp_align) idle = false;                                   // (only the stream assignment below looks at it from here on)
    // Launches big enough to crowd each other out start on the handle's OWN stream, whatever that costs at the batch's end.  Aligning
    // the round robin to end there puts solve 0 of an even batch on the internal stream -- a queue that has been idle since the last
    // batch and wakes up later than the handle's, which has just carried the caller's work: solve 1 was then dispatched BEFORE solve 0,
    // its waiting workgroups took the slots solve 0 needed, and the bounded waits expired (round 4, tools/_seq tests: 3-7 of 18 fresh
    // 64-instance handles with even batch lengths, none with odd ones; it is what made test_big_batches_... fail one run in eight).
    const bool crowd = 2 * (size_t)h->p.B * (h->p.nblk + 1) > slots;
    if (crowd && !exp_env("BN_ALIGN_BIG")) idle = false;      // (BN_ALIGN_BIG: tools/recovery_stress.py brings the race back)
    int rc = BN_OK;
    int stamp = 0;
    for (int32_t i = 0; i < n && rc == BN_OK; ++i) {
        const float *e = eps ? eps + (size_t)(i % eps_ring) * (size_t)eps_stride : nullptr;
        // Stream (and trajectory buffer) of solve i: round robin, except that the LAST solve always runs on the handle's own stream
        // -- whatever follows it there (the batch's tail kernel, a getter) is then ordered by the queue itself instead of by a
        // cross-queue event that has only just fired (measured: 18 us between the last rollout kernel and the tail).  With an
        // even n the last two solves share the stream, which costs one overlap.  (Not the first two: the solve on the other
        // stream would then become eligible before its predecessor is even dispatched -- see the note on eligibility above.)
        // On an idle stream nothing ties the first solve to the handle's stream, so the round robin is simply aligned to END there.
        const int q = idle ? (n - 1 - i) % S : (i == n - 1 ? 0 : i % S);
        // A long batch ends with its own tail (see below): on the latency kernel it rides in the last launch as a second aux workgroup.
        const bool own_tail = i == n - 1 && n >= kEagerTailMinBatch && exp_self_tail;
        // A launch that exceeds one residency round (`over` below; up to 1.25 rounds overlap at all, see `mine`): a launch whose workgroups
        // wait for partials holds slots, and the dispatcher does not share freed slots fairly between two queues -- a successor that
        // becomes eligible while its predecessor still has workgroups to place can take every slot and starve it until the bounded waits
        // expire.  In the steady state that cannot happen: launch i+1 becomes eligible when launch i-1 completes, a whole kernel after
        // launch i did.  The START of a batch is different -- the second launch goes to a queue that has been idle and wakes up late
        // (8 us and more in the kernel traces): it could be dispatched before the first (the inversion the stream assignment above made
        // rare), or so late that the THIRD launch, eligible when the first completes, finds it half placed (round 4,
        // tools/fuzz_features.py: every expiry on a 70-instance planner was the third launch of a batch waiting for workgroups of the
        // second that never got a slot: one handle in 20 to 400, box by box, each repaired by a re-run).  So the first three launches
        // are handed over one by one: a one-thread marker kernel in front of a launch stamps pinned host memory when its queue has
        // reached it, and the host enqueues the next launch only then -- the other queue is woken by a marker of its own right away.
        // A few microseconds of host time under a first launch of 20 us and more, once per batch.  (The stamps cannot come from the
        // rollout kernels themselves: two more live scalars in their prologues put the role kernel on the register allocator's
        // emergency slot, tests/test_build_artifacts.py.)
        const hipStream_t qs = q ? h->xstream[q - 1] : h->stream;
        auto seen = [&](int word, hipStream_t wait_out) -> int {
            const auto t0 = std::chrono::steady_clock::now();
            while (__atomic_load_n(&h->h_err[word], __ATOMIC_ACQUIRE) != stamp) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {      // something else holds the GPU: wait the launch out
                    BN_HIP(hipStreamSynchronize(wait_out));
                    break;
                }
            }
            return BN_OK;
        };
        // Only launches that exceed one residency round by THEMSELVES can be starved for good (a waiting launch of at most one round
        // leaves its predecessor's running workgroups to finish and free what the rest of it needs): two launches that merely do not
        // fit side by side -- 256 instances on the one-wave kernel, K = 16384 -- skip the hand-over.
        const bool over = (size_t)h->p.B * (h->p.nblk + 1) > slots;
        if (over && !h->replaying && n >= 3 && i < 3) {
            if (i == 0) {
                stamp = (int)(h->solves & 0x3fffffff) + 1;
                for (int w = 4; w < 8; ++w) __atomic_store_n(&h->h_err[w], 0, __ATOMIC_RELEASE);
                if (S > 1) BN_HIP(bn::launch_stamp(h->d_err + 5, stamp, h->xstream[0]));      // wakes the other queue
                BN_HIP(bn::launch_stamp(h->d_err + 4, stamp, qs));
            } else if (i == 1) {
                if (int r = seen(4, h->stream)) return r;                                          // the first launch is next in a queue that is being processed
                if (S > 1) { if (int r = seen(5, h->xstream[0])) return r; }
                BN_HIP(bn::launch_stamp(h->d_err + 6, stamp, qs));
            } else {
                if (int r = seen(6, h->xstream[0])) return r;                                      // ... and so is the second
            }
        }
        rc = solve_impl(h, states, states_where, e, noise, false, true, qs, q, own_tail);
        if ((i & 63) == 63) tune_sample(h);
    }
    // A long batch ends with its own tail, enqueued right behind the last solve and BEFORE the join: a tail kernel that comes later
    // (flush, sync, a getter) would sit behind the join's barrier packet, ~10 us of queue processing after the last rollout kernel.
    // Short batches keep the tail pending: chained short batches carry it in their next launch for free.
    if (rc == BN_OK && n >= kEagerTailMinBatch && !h->in_episode) rc = flush_tail(h);
    static const bool exp_join = exp_env("BN_JOIN") != nullptr;   // experiments (tools/region_overhead.py)
    for (int q = 0; exp_join && q + 1 < S; ++q) {                   // join: the handle's stream continues behind all of them
        hipError_t e1 = hipEventRecord(h->ev_join[q], h->xstream[q]);
        hipError_t e2 = hipStreamWaitEvent(h->stream, h->ev_join[q], 0);
        if (rc == BN_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(BN_ERR_HIP, "joining the overlapped launches failed");
    }
    if (rc == BN_OK && !h->in_episode) journal_push(h, rec, replayable);
    return rc;
}

int bn_mppi_env_attach(bn_mppi_t *h, const float *latent_mean, const float *latent_std, bn_mem_kind where,
                       float goal_threshold, float delta_t, uint64_t seed)
{
    if (!h || !latent_mean || !latent_std) return fail(BN_ERR_INVALID, "null argument");
    if (!(goal_threshold >= 0.0f) || !(delta_t > 0.0f)) return fail(BN_ERR_INVALID, "goal_threshold >= 0 and delta_t > 0 required");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t bytes = (size_t)h->n_maps * h->p.G * h->p.G * 4;
    if (!h->d_lat_mean) {
        BN_HIP(hipMalloc((void **)&h->d_lat_mean, bytes));
        BN_HIP(hipMalloc((void **)&h->d_lat_std, bytes));
        BN_HIP(hipMalloc((void **)&h->d_env_state, (size_t)h->p.B * 3 * 4));
        BN_HIP(hipMalloc((void **)&h->d_ep_done, (size_t)h->p.B * sizeof(int)));
    }
    const hipMemcpyKind kind = where == BN_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    BN_HIP(hipMemcpy(h->d_lat_mean, latent_mean, bytes, kind));
    BN_HIP(hipMemcpy(h->d_lat_std, latent_std, bytes, kind));
    h->p.lat_mean = h->d_lat_mean; h->p.lat_std = h->d_lat_std; h->p.env_state = h->d_env_state; h->p.ep_done = h->d_ep_done;
    h->p.goal_thr = goal_threshold; h->p.env_dt = delta_t; h->p.env_seed = seed;
    // latency kernel, overlapped episodes: how far one environment step can move the start cell (see SolveParams::spec_extra)
    h->p.spec_extra = 0;
    if (h->lat_kernel && h->d_gran[0] && !exp_env("BN_NO_SPEC_WINDOW")) {
        const double vmax = std::max(std::fabs((double)h->p.umin0), std::fabs((double)h->p.umax0));
        const double cells = std::floor(vmax * (double)delta_t / (double)h->p.res) + 1.0;
        if (cells <= 8.0 && h->p.WN + 2 * (int)cells <= h->p.G + 1) {
            h->p.spec_extra = (int)cells;
            if (bn::lat_lds_bytes(h->p) == 0) h->p.spec_extra = 0;      // would not fit the LDS
        }
    }
    h->env_attached = true;
    return BN_OK;
}

int bn_mppi_env_set_freeze(bn_mppi_t *h, int32_t freeze_on_goal)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    h->p.env_freeze = freeze_on_goal ? 1 : 0;
    return BN_OK;
}

int bn_mppi_env_step(bn_mppi_t *h, const float *actions_device, float *states_device, float *rewards_device,
                     int32_t *terminated_device, const float *z_device, uint64_t step_index)
{
    if (!h || !actions_device || !states_device || !rewards_device || !terminated_device) return fail(BN_ERR_INVALID, "null argument");
    if (!h->env_attached) return fail(BN_ERR_STATE, "bn_mppi_env_attach must precede bn_mppi_env_step");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = guard_foreign_overlap(h)) return rc;
    BN_HIP(bn::launch_env_step(h->p, actions_device, states_device, rewards_device, terminated_device, z_device, step_index, h->stream));
    return BN_OK;
}

int bn_mppi_env_collision_check(bn_mppi_t *h, const float *states_device, int32_t n_positions, float stuck_threshold,
                                const float *z_device, uint64_t draw_index, uint8_t *out_device)
{
    if (!h || !states_device || !out_device) return fail(BN_ERR_INVALID, "null argument");
    if (n_positions < 1) return fail(BN_ERR_INVALID, "n_positions must be >= 1");
    if (!h->env_attached) return fail(BN_ERR_STATE, "bn_mppi_env_attach must precede bn_mppi_env_collision_check");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = guard_foreign_overlap(h)) return rc;
    BN_HIP(bn::launch_env_collision(h->p, states_device, n_positions, stuck_threshold, z_device, draw_index, out_device, h->stream));
    return BN_OK;
}

int bn_mppi_episode_async(bn_mppi_t *h, int32_t n_steps, const float *states0, bn_mem_kind states_where, const float *eps,
                          bn_noise_kind noise, int32_t eps_ring, int64_t eps_stride, const float *z_device)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->env_attached) return fail(BN_ERR_STATE, "bn_mppi_env_attach must precede bn_mppi_episode_async");
    if (h->slow_path) return fail(BN_ERR_INVALID, "the fused device-side closed loop is not available on the slow path (a horizon beyond the role kernels' LDS, or the "
                                                   "reference-order arithmetic together with sampled slip): drive bn_mppi_forward_async + bn_mppi_env_step instead");
    if (!h->pipelined) return fail(BN_ERR_INVALID, "the device-side closed loop needs the pipelined mode (num_samples <= 4096)");
    if (n_steps < 1) return fail(BN_ERR_INVALID, "n_steps must be >= 1");
    BN_BIND(h);
    if (int rc = flush_tail(h)) return rc;            // anything pending belongs to the pre-episode state
    const size_t B = h->p.B;
    if (n_steps > h->ep_steps) {
        BN_HIP(hipStreamSynchronize(h->stream));
        if (h->d_ep_states) BN_HIP(hipFree(h->d_ep_states));
        if (h->d_ep_reward) BN_HIP(hipFree(h->d_ep_reward));
        if (h->d_ep_action) BN_HIP(hipFree(h->d_ep_action));
        h->d_ep_states = h->d_ep_reward = h->d_ep_action = nullptr;
        BN_HIP(hipMalloc((void **)&h->d_ep_states, (size_t)(n_steps + 1) * B * 3 * 4));
        BN_HIP(hipMalloc((void **)&h->d_ep_reward, (size_t)n_steps * B * 4));
        BN_HIP(hipMalloc((void **)&h->d_ep_action, (size_t)n_steps * B * 2 * 4));
        h->ep_steps = n_steps;
    }
    h->p.ep_states = h->d_ep_states; h->p.ep_reward = h->d_ep_reward; h->p.ep_action = h->d_ep_action;
    BN_HIP(hipMemsetAsync(h->d_ep_done, 0xff, B * sizeof(int), h->stream));        // -1: goal not reached
    if (!states0) return fail(BN_ERR_INVALID, "states0 is null");
    if (states_where == BN_MEM_HOST) {                 // one upload; the loop below must not touch the host again
        BN_HIP(hipStreamSynchronize(h->stream));
        BN_HIP(hipMemcpy(h->d_state, states0, B * 3 * 4, hipMemcpyHostToDevice));
        states0 = h->d_state;
        states_where = BN_MEM_DEVICE;
    }
    journal_arm_states(h);                            // the first step's launch keeps states0 for a re-run
    h->in_episode = true;
    h->ep_len = 0;
    h->ep_z = z_device;
    int rc = bn_mppi_solve_n_async(h, n_steps, states0, states_where, eps, noise, eps_ring, eps_stride);
    if (rc == BN_OK) rc = flush_tail(h);              // the last solve's tail and the last environment step
    h->in_episode = false;
    if (rc == BN_OK) journal_push(h, bn_mppi::BatchRec{n_steps, states0, eps, noise, eps_ring, eps_stride, true, z_device}, noise != BN_NOISE_HOST_KT2);
    return rc;
}

int bn_mppi_episode_log(bn_mppi_t *h, float *states_host, float *rewards_host, float *actions_host, int32_t *done_step_host)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!h->d_ep_states || h->ep_len < 1) return fail(BN_ERR_STATE, "no episode has been run");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t B = h->p.B, n = (size_t)h->ep_len;
    if (states_host) BN_HIP(hipMemcpy(states_host, h->d_ep_states, (n + 1) * B * 3 * 4, hipMemcpyDeviceToHost));
    if (rewards_host) BN_HIP(hipMemcpy(rewards_host, h->d_ep_reward, n * B * 4, hipMemcpyDeviceToHost));
    if (actions_host) BN_HIP(hipMemcpy(actions_host, h->d_ep_action, n * B * 2 * 4, hipMemcpyDeviceToHost));
    if (done_step_host) BN_HIP(hipMemcpy(done_step_host, h->d_ep_done, B * sizeof(int), hipMemcpyDeviceToHost));
    return BN_OK;
}

int bn_mppi_dwa_solve(bn_mppi_t *h, const float *states_host, const float *actions_host, int32_t num_actions,
                      const float *stage_goal_host, float *best_action_host, float *best_states_host,
                      float *costs_host, float *weights_host, float *states_all_host, int32_t *best_index_host)
{
    if (!h || !states_host || !actions_host) return fail(BN_ERR_INVALID, "null argument");
    if (num_actions < 1 || num_actions > 1024) return fail(BN_ERR_INVALID, "num_actions must be in [1, 1024]");
    if (!h->map_set || !h->goal_set) return fail(BN_ERR_STATE, "set_map and set_goal must precede dwa_solve");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    const size_t B = h->p.B, NA = num_actions, T1 = h->p.T + 1;
    // scratch layout: actions | stage goal | X | cost | w | best | best states | states
    const size_t n_act = B * NA * 2, n_goal = B * 2, n_X = B * NA * T1 * 3, n_c = B * NA, n_bs = B * T1 * 3;
    const size_t floats = n_act + n_goal + n_X + 2 * n_c + B + n_bs + B * 3;
    if (int rc = ensure_scratch(h, floats * 4)) return rc;
    float *d_act = h->d_scratch, *d_goal = d_act + n_act, *d_X = d_goal + n_goal, *d_c = d_X + n_X, *d_w = d_c + n_c;
    int *d_best = reinterpret_cast<int *>(d_w + n_c);
    float *d_bs = d_w + n_c + B, *d_st = d_bs + n_bs;
    // one staged upload (actions | stage goal) + the states, one launch, the downloads queued behind it, one wait
    h->dwa_stage.resize(n_act + n_goal);
    std::memcpy(h->dwa_stage.data(), actions_host, n_act * 4);
    if (stage_goal_host) std::memcpy(h->dwa_stage.data() + n_act, stage_goal_host, n_goal * 4);
    BN_HIP(hipStreamSynchronize(h->stream));            // the staging vector and the scratch may still feed a previous call
    BN_HIP(hipMemcpyAsync(d_act, h->dwa_stage.data(), (n_act + (stage_goal_host ? n_goal : 0)) * 4, hipMemcpyHostToDevice, h->stream));
    if (!stage_goal_host) BN_HIP(hipMemcpyAsync(d_goal, h->d_goal, n_goal * 4, hipMemcpyDeviceToDevice, h->stream));   // no reference path: the goal (dwa.py:243-247)
    BN_HIP(hipMemcpyAsync(d_st, states_host, B * 3 * 4, hipMemcpyHostToDevice, h->stream));
    bn::SolveParams p = h->p;
    p.state = d_st;
    BN_HIP(bn::launch_dwa(p, d_act, d_goal, num_actions, d_X, d_c, d_w, d_best, d_bs, nullptr, h->stream));
    std::vector<int> best(B);
    BN_HIP(hipMemcpyAsync(best.data(), d_best, B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (best_states_host) BN_HIP(hipMemcpyAsync(best_states_host, d_bs, n_bs * 4, hipMemcpyDeviceToHost, h->stream));
    if (costs_host) BN_HIP(hipMemcpyAsync(costs_host, d_c, n_c * 4, hipMemcpyDeviceToHost, h->stream));
    if (weights_host) BN_HIP(hipMemcpyAsync(weights_host, d_w, n_c * 4, hipMemcpyDeviceToHost, h->stream));
    if (states_all_host) BN_HIP(hipMemcpyAsync(states_all_host, d_X, n_X * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    for (size_t b = 0; b < B; ++b) {
        if (best_index_host) best_index_host[b] = best[b];
        if (best_action_host) std::memcpy(best_action_host + b * 2, actions_host + (b * NA + best[b]) * 2, 8);
    }
    return BN_OK;
}

int bn_mppi_dwa_forward_async(bn_mppi_t *h, const float *states_device, float *prev_action_device, const float a_lim_host[2],
                              float dwa_delta_t, int32_t num_lin_vel, int32_t num_ang_vel, const float *path_device, int32_t num_path,
                              float lookahead, float *best_states_device)
{
    if (!h || !states_device || !prev_action_device || !a_lim_host) return fail(BN_ERR_INVALID, "null argument");
    const int64_t NA = (int64_t)num_lin_vel * num_ang_vel;
    if (num_lin_vel < 1 || num_ang_vel < 1 || NA > 1024) return fail(BN_ERR_INVALID, "num_lin_vel * num_ang_vel must be in [1, 1024]");
    if (path_device && num_path < 1) return fail(BN_ERR_INVALID, "a reference path needs at least one point");
    if (!h->map_set || !h->goal_set) return fail(BN_ERR_STATE, "set_map and set_goal must precede dwa_forward");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    const size_t B = h->p.B, T1 = h->p.T + 1;
    // the scratch layout of bn_mppi_dwa_solve: actions | stage goal | X | cost | w | best | best states | states
    const size_t n_act = B * NA * 2, n_goal = B * 2, n_X = B * NA * T1 * 3, n_c = B * NA, n_bs = B * T1 * 3;
    const size_t floats = n_act + n_goal + n_X + 2 * n_c + B + n_bs + B * 3;
    if (floats * 4 > h->scratch_bytes) BN_HIP(hipStreamSynchronize(h->stream));      // growing frees the old block: nothing may still use it
    if (int rc = ensure_scratch(h, floats * 4)) return rc;
    float *d_act = h->d_scratch, *d_goal = d_act + n_act, *d_X = d_goal + n_goal, *d_c = d_X + n_X, *d_w = d_c + n_c;
    int *d_best = reinterpret_cast<int *>(d_w + n_c);
    float *d_bs = best_states_device ? best_states_device : d_w + n_c + B;
    bn::SolveParams p = h->p;
    p.state = states_device;
    BN_HIP(bn::launch_dwa_window(p, prev_action_device, a_lim_host, dwa_delta_t, num_lin_vel, num_ang_vel, path_device, num_path, lookahead,
                                 d_act, d_goal, h->stream));
    BN_HIP(bn::launch_dwa(p, d_act, d_goal, (int)NA, d_X, d_c, d_w, d_best, d_bs, prev_action_device, h->stream));
    return BN_OK;
}

int bn_mppi_dwa_buffers(bn_mppi_t *h, int32_t num_actions, const float **states_all_device, const float **costs_device,
                        const float **weights_device)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (num_actions < 1 || num_actions > 1024) return fail(BN_ERR_INVALID, "num_actions must be in [1, 1024]");
    const size_t B = h->p.B, NA = num_actions, T1 = h->p.T + 1;
    const size_t n_act = B * NA * 2, n_goal = B * 2, n_X = B * NA * T1 * 3, n_c = B * NA;
    if (!h->d_scratch || h->scratch_bytes < (n_act + n_goal + n_X + 2 * n_c + B + B * T1 * 3 + B * 3) * 4)
        return fail(BN_ERR_STATE, "no bn_mppi_dwa_solve with this num_actions has run");
    const float *d_X = h->d_scratch + n_act + n_goal;          // the scratch layout of bn_mppi_dwa_solve
    if (states_all_device) *states_all_device = d_X;
    if (costs_device) *costs_device = d_X + n_X;
    if (weights_device) *weights_device = d_X + n_X + n_c;
    return BN_OK;
}

int bn_mppi_dwa_candidates(bn_mppi_t *h, int32_t num_actions, const float **actions_device, const float **stage_goal_device)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (num_actions < 1 || num_actions > 1024) return fail(BN_ERR_INVALID, "num_actions must be in [1, 1024]");
    if (!h->d_scratch) return fail(BN_ERR_STATE, "no DWA solve has run");
    if (actions_device) *actions_device = h->d_scratch;                                                    // (B, num_actions, 2)
    if (stage_goal_device) *stage_goal_device = h->d_scratch + (size_t)h->p.B * num_actions * 2;            // (B, 2)
    return BN_OK;
}

int bn_mppi_sync(bn_mppi_t *h)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);
    hp_cancel(h);
    if (int rc = flush_tail(h)) return rc;
    if (int rc = sync_checked(h)) return rc;            // a bounded device-side wait of an overlapped launch may have expired: see recover_overlap
    return self_check(h);
}

int bn_mppi_order_outputs(bn_mppi_t *h)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);                                         // (that is the call: see hp_order_now)
    return BN_OK;
}

uint64_t bn_mppi_recovery_count(const bn_mppi_t *h) { return h ? h->recoveries : 0; }

int32_t bn_mppi_overlap_mode(const bn_mppi_t *h)
{
    if (!h) return -1;
    const bool capable = (h->lat_kernel || h->role_overlap || h->ticket_overlap) && h->n_streams > 1 && !(h->cfg.flags & (BN_FLAG_PROFILE | BN_FLAG_NO_OVERLAP));
    if (!capable) return 3;
    if (h->overlap_off) return 2;
    return h->tune.chosen ? 1 : 0;
}

int bn_mppi_debug_cadence(bn_mppi_t *h, int32_t mode, double us_per_launch)
{
    if (!h || mode < 0 || mode > 1 || !(us_per_launch > 0)) return fail(BN_ERR_INVALID, "bad argument");
    tune_record(h, mode, us_per_launch);
    return h->tune.explore ? 1 : 0;
}

int bn_mppi_first_action(bn_mppi_t *h, int32_t instance, float action_host[2])
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!action_host) return fail(BN_ERR_INVALID, "null output");
    if (h->solves == 0) return fail(BN_ERR_STATE, "no solve has run");
    if (h->shard_pending) return fail(BN_ERR_STATE, "a sharded solve waits for bn_mppi_shard_finish_async");
    BN_BIND_Q(h);                                           // (the mailbox lives on the host: nothing of the handle's stream is touched, nothing to order)
    if (int rc = flush_tail(h)) return rc;                  // the tail is what posts it
    // The tail writes U*[0] to pinned host memory as soon as its merge is done -- two {value, tag} granules, tag = solve index + 1
    // -- and goes on with X* and the weights; the host polls the granules: no stream synchronisation, no copy.
    uint32_t want = (uint32_t)(h->solves - (h->hp_armed ? 1 : 0));      // (a launch that waits for the NEXT state is not the latest solve)
    const volatile unsigned long long *m = h->h_mail + 2 * (size_t)instance;
    for (long it = 0;; ++it) {
        if ((it & 0xff) == 0xff && h->hp_posted_tag && hp_gave_up(h, h->hp_posted_tag)) {      // host-paced: the request went to a launch that had just given up
            if (int rc = hp_redo(h)) return rc;
            want = (uint32_t)(h->solves - (h->hp_armed ? 1 : 0));
        }

        const unsigned long long a = m[0], b = m[1];
        if ((uint32_t)(a >> 32) == want && (uint32_t)(b >> 32) == want) {
            const uint32_t ua = (uint32_t)a, ub = (uint32_t)b;
            std::memcpy(action_host, &ua, 4); std::memcpy(action_host + 1, &ub, 4);
            break;
        }
        if ((it & 0xfff) == 0xfff) {                        // every few microseconds: is the stream still working on it?
            const hipError_t q = hipStreamQuery(h->stream);
            (void)hipGetLastError();
            if (q == hipSuccess && it > (1L << 22)) return fail(BN_ERR_HIP, "the stream is idle but the first action of solve %llu never arrived",
                                                                 (unsigned long long)h->solves - 1);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(BN_ERR_HIP, "the stream failed while waiting for the first action");
        }
    }
    if (int rc = self_check(h)) return rc;                  // (a one-launch solve: its tail's wait ended before it posted)
    if (h->overlap_used && __atomic_load_n(h->h_err, __ATOMIC_ACQUIRE)) {   // an expired wait upstream: repair, then read the repaired value
        if (int rc = settle_point(h)) return rc;
        const unsigned long long a = m[0], b = m[1];
        const uint32_t ua = (uint32_t)a, ub = (uint32_t)b;
        std::memcpy(action_host, &ua, 4); std::memcpy(action_host + 1, &ub, 4);
    }
    return BN_OK;
}

int bn_mppi_debug_expire_wait(bn_mppi_t *h)
{
    // Test hook: behave as if a wait had expired in the batches enqueued since the last synchronisation point -- the error word is
    // set and everything those batches wrote (mean, U* | X*, weights, costs, trajectories) is overwritten with NaN patterns, so a
    // caller that gets valid results afterwards got them from the re-run.
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);
    hp_cancel(h);
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    const size_t B = h->p.B, K = h->p.K, T = h->p.T;
    BN_HIP(hipMemset(h->d_mean, 0xff, B * T * 2 * 4));
    BN_HIP(hipMemset(h->d_ustar, 0xff, (B * T * 2 + B * (T + 1) * 3) * 4));
    BN_HIP(hipMemset(h->d_w, 0xff, B * K * 4));
    BN_HIP(hipMemset(h->d_cost_out, 0xff, B * K * 4));
    if (h->d_X) BN_HIP(hipMemset(h->d_X, 0xff, B * (T + 1) * 3 * (size_t)h->p.Kp * 4));
    if (h->d_ticket) BN_HIP(hipMemset(h->d_ticket, 0x01, B * 65 * 4));      // what two launches drawing tickets at once leave behind (recover_overlap)
    if (h->d_ep_states && h->ep_len > 0) {
        BN_HIP(hipMemset(h->d_ep_states, 0xff, (size_t)(h->ep_len + 1) * B * 3 * 4));
        BN_HIP(hipMemset(h->d_ep_reward, 0xff, (size_t)h->ep_len * B * 4));
        BN_HIP(hipMemset(h->d_ep_action, 0xff, (size_t)h->ep_len * B * 2 * 4));
    }
    *h->h_err = 1;
    h->overlap_used = true;
    return BN_OK;
}

int bn_mppi_debug_host_paced(bn_mppi_t *h, int32_t polls, int32_t post_unchecked)
{
    // Test hook: how many looks (~2 us each) a prelaunched solve waits for its state before it gives up (default 25000), and whether the
    // host posts without checking that the launch is still there -- with one look the launch is gone before any host can answer, and
    // an unchecked post lands in the repair path of bn_mppi_first_action.
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (polls < 1) return fail(BN_ERR_INVALID, "polls must be >= 1");
    h->hp_polls = polls;
    h->hp_skip_check = post_unchecked != 0;
    return BN_OK;
}

int bn_mppi_flush(bn_mppi_t *h)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);
    hp_cancel(h);
    h->hp_gran_valid = false;                          // (the caller may write the mean buffer next: the class's _previous_action_seq setter)
    if (int rc = flush_tail(h)) return rc;
    // no synchronisation here; but an expiry that has ALREADY happened is repaired now rather than at the next synchronising call
    return (h->overlap_used && !h->replaying && __atomic_load_n(h->h_err, __ATOMIC_ACQUIRE)) ? settle_overlap(h, false) : BN_OK;
}

int bn_mppi_solve(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps,
                  bn_noise_kind noise, float *ustar_host, float *xstar_host)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (h->lat_kernel && h->tail_pending && !h->self_off) { if (int rc = flush_tail(h)) return rc; }      // (as in forward_impl)
    if (int rc = solve_impl(h, states, states_where, eps, noise, false, false, nullptr, 0, true)) return rc;   // own tail in the launch where that exists
    if (int rc = flush_tail(h)) return rc;
    const size_t B = h->p.B, T = h->p.T;
    if (ustar_host) BN_HIP(hipMemcpyAsync(ustar_host, h->d_ustar, B * T * 2 * 4, hipMemcpyDeviceToHost, h->stream));
    if (xstar_host)
        BN_HIP(hipMemcpyAsync(xstar_host, h->d_xstar, B * (T + 1) * 3 * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    return self_check(h);
}

// Rows of the latest solve's trajectory batch, regenerated (launch_reroll).  The tail of that solve must have run
// (mean_used, state copy); the caller's eps block of that solve must still be alive when the noise was injected.
static int reroll_rows(bn_mppi_t *h, int32_t instance, const int *idx_device, int32_t n, float *out_device)
{
    if (h->p.slip_on) return fail(BN_ERR_INVALID, "re-rolling is not available in sampled-slip mode");
    if (h->solves == 0) return fail(BN_ERR_STATE, "no solve has run");
    if (h->shard_pending) return fail(BN_ERR_STATE, "a sharded solve waits for bn_mppi_shard_finish_async");
    // The rows are rolled out again on the map as it is NOW: after a bn_mppi_set_map they would not be the latest solve's rollouts
    // (the reference's get_top_samples returns the batch forward() stored, mppi.py:221-240) -- refuse instead of answering for another map.
    if (h->map_epoch != h->map_epoch_at_solve)
        return fail(BN_ERR_STATE, "the map changed since the latest solve: its rollouts cannot be regenerated (read them before bn_mppi_set_map, or solve again)");
    if (int rc = flush_tail(h)) return rc;
    bn::SolveParams p = h->p;
    const int cur = (int)((h->solves - 1) % kSlots);
    p.solve = h->solves - 1;
    p.state = h->d_state_copy[cur];
    p.eps = h->last_eps;
    if (int rc = guard_foreign_overlap(h)) return rc;
    BN_HIP(bn::launch_reroll(p, h->last_mode, instance, idx_device, n, out_device, h->stream));
    return BN_OK;
}

int bn_mppi_reroll_async(bn_mppi_t *h, int32_t instance, const int32_t *idx_device, int32_t n, float *out_device)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (n < 0 || (!idx_device && n > h->p.K)) return fail(BN_ERR_INVALID, "n=%d out of range", n);
    if (n == 0) return BN_OK;
    if (!out_device) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    return reroll_rows(h, instance, idx_device, n, out_device);
}

static int copy_out(bn_mppi_t *h, int32_t instance, const float *dev, size_t per_instance, float *out_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!out_host) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    BN_HIP(hipMemcpy(out_host, dev + (size_t)instance * per_instance, per_instance * 4, hipMemcpyDeviceToHost));
    return BN_OK;
}

int bn_mppi_get_weights(bn_mppi_t *h, int32_t instance, float *out_host)
{
    return h ? copy_out(h, instance, h->d_w, h->p.K, out_host) : fail(BN_ERR_INVALID, "null handle");
}

int bn_mppi_get_costs(bn_mppi_t *h, int32_t instance, float *out_host)
{
    return h ? copy_out(h, instance, h->d_cost_out, h->p.K, out_host) : fail(BN_ERR_INVALID, "null handle");
}

int bn_mppi_get_states(bn_mppi_t *h, int32_t instance, float *out_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!out_host) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t K = h->p.K, Kp = h->p.Kp, T1 = h->p.T + 1, n = K * T1 * 3;
    if (int rc = ensure_scratch(h, n * 4)) return rc;
    if (h->p.lean) {                                   // not materialised: regenerate all K rows
        if (int rc = reroll_rows(h, instance, nullptr, (int32_t)K, h->d_scratch)) return rc;
    } else
    BN_HIP(bn::launch_states_to_reference((h->x_idx && h->d_Xalt[0] ? h->d_Xalt[0] : h->d_X) + (size_t)instance * Kp * T1 * 3, h->d_scratch, (int)K, (int)Kp, (int)T1,
                                          h->stream));
    BN_HIP(hipMemcpyAsync(out_host, h->d_scratch, n * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    return BN_OK;
}

int bn_mppi_get_controls(bn_mppi_t *h, int32_t instance, float *out_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!out_host) return fail(BN_ERR_INVALID, "null output");
    if (!h->d_U || !h->p.store_u) return fail(BN_ERR_STATE, "controls are only stored with BN_FLAG_STORE_CONTROLS");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t K = h->p.K, Kp = h->p.Kp, T = h->p.T, n = K * T * 2;
    if (int rc = ensure_scratch(h, n * 4)) return rc;
    BN_HIP(bn::launch_controls_to_reference((h->x_idx && h->d_Ualt[0] ? h->d_Ualt[0] : h->d_U) + (size_t)instance * Kp * T * 2, h->d_scratch, (int)K, (int)Kp, (int)T,
                                            h->stream));
    BN_HIP(hipMemcpyAsync(out_host, h->d_scratch, n * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    return BN_OK;
}

int bn_mppi_get_philox_noise(bn_mppi_t *h, int32_t instance, uint64_t solve_index, float *out_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!out_host) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t n = (size_t)h->p.K * h->p.T * 2;
    if (int rc = ensure_scratch(h, n * 4)) return rc;
    BN_HIP(bn::launch_philox_noise(h->d_scratch, h->p.seed, solve_index, instance, h->p.K, h->p.T, h->p.k0, h->stream));
    BN_HIP(hipMemcpyAsync(out_host, h->d_scratch, n * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    return BN_OK;
}

int bn_mppi_get_slip_noise(bn_mppi_t *h, int32_t instance, uint64_t solve_index, float *zt_host, float *zc_host, float *zo_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (!zt_host || !zc_host || !zo_host) return fail(BN_ERR_INVALID, "null output");
    if (!h->p.slip_on) return fail(BN_ERR_STATE, "handle was created without BN_FLAG_SAMPLED_SLIP");
    BN_BIND(h);
    const size_t K = h->p.K, T = h->p.T, nt = K * T, nc = K * (T + 1);
    if (int rc = ensure_scratch(h, (nt + nc + T) * 4)) return rc;
    float *zt = h->d_scratch, *zc = zt + nt, *zo = zc + nc;
    BN_HIP(bn::launch_philox_slip(zt, zc, zo, h->p.seed, solve_index, instance, (int)K, (int)T, h->stream));
    BN_HIP(hipMemcpyAsync(zt_host, zt, nt * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipMemcpyAsync(zc_host, zc, nc * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipMemcpyAsync(zo_host, zo, T * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    return BN_OK;
}

int bn_mppi_get_top_samples(bn_mppi_t *h, int32_t instance, int32_t n, float *states_host, float *weights_host)
{
    if (int rc = check_instance(h, instance, false)) return rc;
    if (n < 0 || n > h->p.K) return fail(BN_ERR_INVALID, "n=%d must satisfy 0 <= n <= K=%d", n, h->p.K);  // mppi.py:229
    if (n == 0) return BN_OK;
    if (!states_host || !weights_host) return fail(BN_ERR_INVALID, "null output");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    const size_t K = h->p.K, T1 = h->p.T + 1;
    std::vector<float> w(K);
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    BN_HIP(hipMemcpy(w.data(), h->d_w + (size_t)instance * K, K * 4, hipMemcpyDeviceToHost));
    std::vector<int> idx(K);
    std::iota(idx.begin(), idx.end(), 0);
    // topk + descending sort (mppi.py:232-238); ties resolved by the lower rollout index
    std::partial_sort(idx.begin(), idx.begin() + n, idx.end(),
                      [&](int a, int b) { return w[a] > w[b] || (w[a] == w[b] && a < b); });
    if ((size_t)n > h->idx_count) {
        if (h->d_idx) BN_HIP(hipFree(h->d_idx));
        h->d_idx = nullptr; h->idx_count = 0;
        BN_HIP(hipMalloc((void **)&h->d_idx, (size_t)n * sizeof(int)));
        h->idx_count = n;
    }
    if (int rc = ensure_scratch(h, (size_t)n * T1 * 3 * 4)) return rc;
    BN_HIP(hipMemcpyAsync(h->d_idx, idx.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (h->p.lean) {                                   // the n winners are re-rolled, bit-identical to a stored batch
        if (int rc = reroll_rows(h, instance, h->d_idx, n, h->d_scratch)) return rc;
    } else
    BN_HIP(bn::launch_gather_states((h->x_idx && h->d_Xalt[0] ? h->d_Xalt[0] : h->d_X) + (size_t)instance * h->p.Kp * T1 * 3, h->d_idx, h->d_scratch, n, h->p.Kp,
                                    (int)T1, h->stream));
    BN_HIP(hipMemcpyAsync(states_host, h->d_scratch, (size_t)n * T1 * 3 * 4, hipMemcpyDeviceToHost, h->stream));
    BN_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) weights_host[i] = w[idx[i]];
    return BN_OK;
}

int bn_mppi_device_buffer(bn_mppi_t *h, bn_buffer_id id, void **device_ptr, size_t *bytes)
{
    if (!h || !device_ptr) return fail(BN_ERR_INVALID, "null argument");
    if (h->shard_tail_inflight) {                     // a library-enqueued sharded solve: order the handle's stream behind its tail first
        BN_BIND(h);
        if (int rc = flush_tail(h)) return rc;
    }
    void *ptrs[BN_BUF_COUNT_] = {h->d_X, h->d_w, h->d_cost_out, h->d_U, h->d_ustar, h->d_xstar, h->d_mean, h->d_map, h->d_goal, h->d_ustar,
                                 h->d_Xalt[0], h->d_Ualt[0]};
    if ((int)id < 0 || id >= BN_BUF_COUNT_) return fail(BN_ERR_INVALID, "unknown buffer id %d", (int)id);
    *device_ptr = ptrs[id];
    if (bytes) *bytes = buffer_bytes(h, id);
    if (!ptrs[id]) return fail(BN_ERR_STATE, "buffer %d is not allocated in this configuration", (int)id);
    return BN_OK;
}

uint64_t bn_mppi_solve_count(const bn_mppi_t *h) { return h ? h->solves - (h->hp_armed ? 1 : 0) : 0; }
int32_t bn_mppi_host_paced(const bn_mppi_t *h) { return h ? (h->hp_enabled ? (h->req_bar ? 2 : 1) : 0) : -1; }
int32_t bn_mppi_states_buffer_index(const bn_mppi_t *h) { return h ? h->x_idx : -1; }

int32_t bn_mppi_arithmetic(const bn_mppi_t *h) { return h ? h->p.ref_order : -1; }
int32_t bn_mppi_fast_quotient(const bn_mppi_t *h) { return h ? (h->p.pow2 ? 2 : h->p.fast_div) : -1; }
int32_t bn_mppi_launches_per_solve(const bn_mppi_t *h) { return h ? ((h->pipelined || h->ticket_mode) ? 1 : 2) : -1; }
int32_t bn_mppi_launches_per_forward(const bn_mppi_t *h)
{
    return h ? ((h->lat_kernel && h->pipelined && !h->self_off && !(h->cfg.flags & BN_FLAG_NO_PIPELINE)) ? 1 : 2) : -1;
}

int32_t bn_mppi_row_pitch(const bn_mppi_t *h) { return h ? h->p.Kp : 0; }

int bn_mppi_kernel_ms(bn_mppi_t *h, float *rollout_ms, float *finish_ms, int32_t *n_solves)
{
    if (!h) return fail(BN_ERR_INVALID, "null handle");
    if (!(h->cfg.flags & BN_FLAG_PROFILE)) return fail(BN_ERR_STATE, "handle was created without BN_FLAG_PROFILE");
    BN_BIND(h);
    if (int rc = settle_point(h)) return rc;
    if (h->pipelined || h->ticket_mode) {
        // close the open group with one more event *before* the flush, then average complete groups only
        const int open = h->prof_in_group;
        if (h->ev_used + 1 > h->ev.size()) {
            hipEvent_t e;
            BN_HIP(hipEventCreate(&e));
            h->ev.push_back(e);
        }
        BN_HIP(hipEventRecord(h->ev[h->ev_used++], h->stream));
        if (int rc = flush_tail(h)) return rc;
        BN_HIP(hipStreamSynchronize(h->stream));
        const size_t marks = h->ev_used;                  // marks-1 intervals; the last one holds `open` launches (or a full group)
        double tot = 0.0;
        size_t launches = 0;
        for (size_t i = 0; i + 1 < marks; ++i) {
            const bool last = (i + 2 == marks);
            const int in_this = (last && open != 0) ? open : kProfGroup;
            if (last && open != 0) break;                 // skip the ragged last group
            float a = 0.0f;
            BN_HIP(hipEventElapsedTime(&a, h->ev[i], h->ev[i + 1]));
            tot += a;
            launches += in_this;
        }
        if (rollout_ms) *rollout_ms = launches ? (float)(tot / launches) : 0.0f;
        if (finish_ms) *finish_ms = 0.0f;
        if (n_solves) *n_solves = (int32_t)launches;
        h->ev_used = 0;
        h->prof_in_group = 0;
        return BN_OK;
    }
    if (int rc = flush_tail(h)) return rc;
    BN_HIP(hipStreamSynchronize(h->stream));
    double r = 0.0, f = 0.0;
    const size_t n = h->ev_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float a = 0.0f, b = 0.0f;
        BN_HIP(hipEventElapsedTime(&a, h->ev[3 * i], h->ev[3 * i + 1]));
        BN_HIP(hipEventElapsedTime(&b, h->ev[3 * i + 1], h->ev[3 * i + 2]));
        r += a; f += b;
    }
    if (rollout_ms) *rollout_ms = n ? (float)(r / n) : 0.0f;
    if (finish_ms) *finish_ms = n ? (float)(f / n) : 0.0f;
    if (n_solves) *n_solves = (int32_t)n;
    h->ev_used = 0;
    return BN_OK;
}

int64_t bn_mppi_algorithmic_bytes(const bn_mppi_t *h, bn_noise_kind noise)
{
    if (!h) return 0;
    const int64_t K = h->p.K, T = h->p.T, G = h->p.G;
    int64_t bytes = 4 * G * G            // risk map read once
                    + (8 * T + 12)       // mean, state
                    + (h->p.lean ? 0 : 12 * K * (T + 1))   // _state_seq_batch write (not in lean mode)
                    + 4 * K              // weights write
                    + 8 * T + 12 * (T + 1);   // U*, X* write
    if (noise != BN_NOISE_PHILOX) bytes += 8 * K * T;   // injected noise read
    if (h->p.store_u) bytes += 8 * K * T;               // _perturbed_action_seqs write
    return bytes;
}

/* The same with the map term replaced by what a solve can reach: the LDS window of (WN x WN) cells around the start state, staged
 * once per instance in the best case (every rollout workgroup stages its own copy; the copies hit in L2).  With BN_FLAG_LEAN the
 * map read is most of SURVEY 8d's lean formula, and a kernel that reads the window only would be credited for bytes it never
 * moves: bench.py reports both fractions. */
int64_t bn_mppi_algorithmic_bytes_window(const bn_mppi_t *h, bn_noise_kind noise)
{
    if (!h) return 0;
    const int64_t G = h->p.G, W = h->p.WN > 0 ? h->p.WN : G;
    return bn_mppi_algorithmic_bytes(h, noise) - 4 * G * G + 4 * W * W;
}

#ifdef BN_TIMING
int bn_mppi_debug_blocks_per_cu(bn_mppi_t *h) { return bn::rollout_blocks_per_cu(h->p); }
/* tools/ablate.py only: device buffer of >= 16 uint64 for the in-kernel cycle stamps */
void bn_mppi_debug_set_stamps(bn_mppi_t *h, void *device_ptr) { h->p.stamps = (unsigned long long *)device_ptr; }
void bn_mppi_debug_trace_by_parity(bn_mppi_t *h, int on) { h->p.trace_by_parity = on; }
#endif

int bn_device_math_eval(int32_t fn, const float *in_device, float *out_device, int64_t n, void *stream)
{
    if (fn < 0 || fn > 5 || !in_device || !out_device || n < 0) return fail(BN_ERR_INVALID, "bad argument");
    if (n == 0) return BN_OK;
    BN_HIP(bn::launch_math_eval(fn, in_device, out_device, (size_t)n, (hipStream_t)stream));
    return BN_OK;
}

const char *bn_last_error(void) { return g_last_error.c_str(); }
int bn_mppi_abi_version(void) { return BN_MPPI_ABI_VERSION; }

}  // extern "C"
