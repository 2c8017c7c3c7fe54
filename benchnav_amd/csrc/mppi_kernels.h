// mppi_kernels.h -- launch interface between the C ABI (mppi_capi.cpp) and the
// gfx950 kernels (mppi_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn {

constexpr int kRolloutsPerBlock = 64;   // lane = rollout; 64 rollouts per workgroup
constexpr int kRolloutThreads = 320;    // 5 wavefronts per workgroup, specialised by role (chain / 2 producers / 2 consumers)
constexpr int kChunk = 4;                // time steps per barrier phase of the rollout kernel (8 was measured slower, round 2)
constexpr int kWaveParkSteps = 30;       // == kParkSteps of wave_park.h (static_assert in rollout_wave.hip)
constexpr int kUPad = 65;               // LDS row pitch of the control tile (bank-conflict-free both ways)
constexpr int kWideFinishThreads = 1024;           // stand-alone tail of the sizes that are not pipelined (K > 2048)
constexpr int kFinishThreads = kRolloutThreads;   // the tail runs as a stand-alone kernel or as the aux workgroup of a rollout launch

enum EpsMode : int { kEpsPhilox = 0, kEpsKT2 = 1, kEpsT2K = 2 };

// Everything the kernels need; passed by value (kernarg segment, scalar loads).
struct SolveParams {
    int K, T, G, B;
    int nblk;            // ceil(K / 64)
    int Kp;              // row pitch of X and U in floats: 64 * nblk (lanes past K store into the pad)
    int WN;              // LDS window edge in cells, 0 = gather from global memory
    int reach;           // ceil(T * vmax * dt / res) + 1 cells
    int spec_extra;      // device-side episodes, latency kernel: cells the state can move in ONE environment step, + 1.  The window is
                         // staged this much wider around the PREVIOUS state while the new one is not known yet; 0 = stage afterwards
    int map_stride;      // G*G if every instance has its own map, 0 if shared
    int pow2;            // resolution is a power of two
    int fast_div;        // general resolution: the in-loop quotient (p - origin) / res (Markstein's three-instruction form, quotient_general
                         // in mppi_device.h) passed the exhaustive check at create
    int store_u;
    int wave_kernel;     // launch the one-wave-per-64-rollouts throughput variant (many workgroups per launch)
    int lat_kernel;      // launch the barrier-free latency variant of the role kernel (every workgroup has a CU to itself)
    int k0;              // global index of this handle's rollout 0 (K-sharded solve: rank r owns rollouts [k0, k0 + K)); keys the Philox stream
    int xs;              // log2 of the instances interleaved along grid x (rollout_grid / decode_wg): 3 keeps the workgroups of
                         // one instance on one XCD (workgroup i runs on XCD i % 8), 0 is instance-per-row
    int aux_prio;        // the aux workgroups run at wave priority 3: set when the launch exceeds one resident round, so that they
                         // start late (in slots freed by the first rollout workgroups) and must not finish last
    int regen_steps;     // one-wave kernel, library noise: non-zero = the epilogue draws the controls it has not parked again; 0 = they take the
                         // round trip through the control buffer (what injected noise always does): VALU against HBM traffic
    int park_steps;      // one-wave kernel: the controls of steps [0, park_steps) wait for their weights in registers of the lane (rollout_wave.inc,
                         // wave_park.h; at most kWaveParkSteps), 0 = the kernel without the register block
    int lds_park;        // one-wave kernel: one more chunk of eight steps waits for its weights in an LDS tile (set when it keeps 16 workgroups per CU)
    int wrap_near;       // reference-order transit: dt * max|omega| < 3 rad, so every heading after the first step's wrap lies where the
                         // branch-free form of the wrap (wrap_angle_near) equals torch.remainder's bit for bit
    int lean;            // lean mode: the (K,T+1,3) trajectory batch is not materialised (bn_mppi_reroll regenerates rows on demand)
    int ref_order;       // BN_FLAG_REFERENCE_ORDER (or dt * max|omega| > 0.5): every transit evaluates sincos_spec of its own heading and
                         // updates in the reference's operation order, x + ((trav v) cos) dt (robot_model.py:86-88) -- the oracle's trig = 2.
                         // Every kernel family has an instantiation in this arithmetic (rollout_role_ref_*.hip, rollout_wave_ref.hip; chain_step<..., REF = true>)
    float res, inv_res;
    float x0, y0;        // index origin == lower clamp (reference grid_map.py:199-201, robot_model.py:93-94)
    float x_hi, y_hi;    // upper clamp
    float dt, thr, lambda_;
    float inv_lambda;           // 1 / lambda_ when that is exact (lambda_ a power of two, reciprocal a normal number), else 0: x / lambda_ == x * inv_lambda bit for bit
    float sigma0, sigma1, iv0, iv1;
    float umin0, umax0, umin1, umax1;
    uint64_t seed;
    uint64_t solve;      // index of this solve in the handle's life = Philox stream position
    int mean_from_part;  // rollout blocks merge part_prev themselves instead of reading `mean`
    int have_prev;       // the launch carries an aux workgroup per instance: tail of the previous solve
    int self_tail;       // ... and a second one per instance: the tail of THIS solve (last launch of an overlapped batch, latency
                         // kernel): it waits on the device for the rollout workgroups of its own launch instead of for a kernel boundary
    const float *map;    // (n_maps, G, G)
    const float *state;  // (B, 3)
    const float *goal;   // (B, 2)
    float *mean;         // (B, T, 2)   read by rollout, rewritten by finish
    float *mean_used;    // (B, T, 2)   lean mode: the mean the latest finished solve sampled around (kept for re-rolls)
    const float *eps;    // per EpsMode, or nullptr
    float *X;            // (B, T+1, 3, Kp)
    float *U;            // (B, T, 2, Kp) or nullptr
    float *cost;         // (B, K)      this solve's per-rollout costs (double-buffered by solve parity)
    float *part;         // (B, nblk, 2 + 2T): block max, block sum, block weighted control sums (double-buffered)
    float *state_copy;   // (B, 3)      the state this solve started from, kept for its tail
    const float *cost_prev, *part_prev, *state_prev;   // the previous solve's buffers (pipelined mode)
    float *cost_out;     // (B, K)      stable copy of the latest finished solve's costs (BN_BUF_COSTS)
    // ---- device-side closed loop ("next" row N2): PlanetaryEnv.step between consecutive solves ----
    int closed_loop;     // rollout workgroups derive this solve's state from the previous one + env step
    int env_on;          // the tail logs the environment step that follows its solve
    int ep_index;        // episode index of the solve whose tail this launch writes = index of the env step the
                         // rollout workgroups apply in closed-loop mode (the step after that solve)
    int env_freeze;      // opt-in (bn_mppi_env_set_freeze): an environment within goal_thr of its goal stays where it is; the
                         // reference keeps stepping a terminated environment (its driver loop stops instead, test_mppi.py:192-194)
    float goal_thr;      // PlanetaryEnv goal_threshold (planetary_env.py:215-217)
    float env_dt;        // PlanetaryEnv delta_t passed to transit (planetary_env.py:203-205)
    uint64_t env_seed;
    const float *lat_mean, *lat_std;   // (n_maps, G, G) latent slip model, observation mode (traversability_model.py:65-69)
    const float *env_z;                // (B,) slip draws of env step `ep_index`, or nullptr (Philox keyed by env_seed)
    float *ep_states;    // (n_steps+1, B, 3) episode log
    float *ep_reward;    // (n_steps, B)      traversability observed by each env step
    float *ep_action;    // (n_steps, B, 2)   the control applied by each env step: U*[0] of that step's solve
    int *ep_done;        // (B)               first step index whose resulting state is within goal_thr, or -1
    float *env_state;    // (B, 3)            latest environment state (after the latest logged step)
    // ---- sampled slip inside the rollouts (BASELINE config 3; reference A9: traversability_model.py:65-69) ----
    int slip_on;         // every traversability lookup draws slip ~ Normal(map, slip_std)[cell]
    const float *slip_std;             // (n_maps, G, G)
    const float *zt, *zc, *zo;         // injected standard normals: transit (B,T,K), cost (B,T+1,K), X* (B,T); or nullptr: Philox
    // fused tail of the sampled kernel: the last workgroup of an instance to finish (ticket) merges the partials itself;
    // weights / X* of that solve are written by the aux workgroup of the NEXT launch, or by the stand-alone tail
    int *ticket;                       // (B, 65) ticket counters (see ticket_merge); reset by the workgroups that draw the last ones
    float *gpart;                      // (B, 64, 2+2T) group rows of the two-level merge (more than 64 workgroups)
    float *ustar_cur, *stats_cur;      // (B, T, 2), (B, 2): merge outputs of this solve (double-buffered by solve parity)
    const float *ustar_prev, *stats_prev;   // merge outputs of the solve whose tail this launch / the stand-alone tail writes
    int tail_merged;                   // the tail reads (ustar_prev, stats_prev) instead of merging `part`
    uint64_t tail_solve;               // index of the solve whose tail is written (its X* draws)
    // ---- overlapped launches (solve_n_overlapped in mppi_capi.cpp): consecutive solves alternate between two streams, so a launch
    // may start while its predecessor still runs; what stream order used to guarantee is carried by monotonic device counters ----
    int overlap;                         // this launch must WAIT on the counters (its predecessor is on the other stream)
    int cur_slot, prev_slot;             // slots (0..2) of this solve's and the previous solve's per-solve buffers
    unsigned long long *flag_part;       // [3][B] rollout workgroups of instance b that have published their partials into slot j, ever
    unsigned long long *flag_tail;       // [B] tails (aux workgroups / finish kernels) of instance b completed, ever
    unsigned long long wait_part;        // flag_part[prev_slot][b] value that means "the previous solve's partials and costs of this instance are all there"
    unsigned long long wait_tail;        // flag_tail[b] value that means "the tail before the one this launch carries is done"
    unsigned long long wait_part_self;   // self_tail: flag_part[cur_slot][b] value that means "this launch's rollout workgroups have all published"
    unsigned long long wait_tail_self;   // self_tail: flag_tail[b] value that means "every earlier tail (incl. the one this launch carries) is done"
    int *err;                            // pinned host memory: set non-zero (system-scope store) when a bounded wait expired
    int *err_dev;                        // the same flag in device memory, for the kernels themselves: writers of `mean` skip it once a wait of the stretch has expired
    float *state_snap;                   // (B, 3) or nullptr: the first launch of a journalled batch keeps the states it was given in memory the handle
                                         // owns -- a re-run must not depend on the caller's buffer still holding them (round 4)
    float *mean_snap;                    // (B, T, 2) or nullptr: workgroup 0 of every instance keeps the mean this solve samples around
                                         // (first launch since the host last checked `err`: where a re-run would start from)
    unsigned long long *gran, *gran_prev;   // (B, nblk, 2+2T) granule copies {value, tag} of this / the previous solve's partial rows
                                            // (overlapped batches with K <= 1024: the successor's prologue polls the rows themselves)
    float *w;            // (B, K)
    float *ustar;        // (B, T, 2)
    float *xstar;        // (B, T+1, 3)
    unsigned long long *mail;   // (B, 2) pinned host memory: the tail posts U*[0] there as two {value, tag} granules the moment the merge is
                                // done -- before the X* rollout and the weights -- for a host that consumes every solve (bn_mppi_first_action)
    float *out_copy;     // optional caller-owned copy of the packed (B,T,2) U* | (B,T+1,3) X* block, written by the same tail
                         // (bn_mppi_forward_async: the drop-in class's fresh output tensors without a second launch)
    float *stats;        // (B, 2): max z, sum exp
    int trace_by_parity;          // timing builds: per-workgroup trace rows of odd solves behind those of even solves (two launches in flight)
    unsigned long long *stamps;   // tools/ablate.py timing builds only (-DBN_TIMING): s_memtime stamps of block 0
    // ---- round 6, appended: the fields above keep the offsets the round-5 kernels were tuned with ----
    int ustar_written;                 // ... and U* itself is in `ustar` already (the K-sharded solve's merge kernel wrote it on the handle's stream:
                                       // a tail on the side stream must not write it again behind the NEXT solve's merge)
    float *out_copy_self;   // ... the same for the tail of THIS launch's solve (self_tail): the aux workgroup of the previous solve's tail
                            // in the same launch must not write the caller's block
    // ---- host-paced launches (rollout_lat.inc HOSTP; bn_mppi_forward_state_async in a loop) ----
    int host_paced;                         // this launch's rollouts start from a state the HOST posts after the launch; sv = the previous state
    const unsigned long long *req_host;     // pinned host memory: kReqGranules {value, tag = solve + 1} granules -- x, y, theta, out pointer lo / hi, command
    unsigned long long *req_dev;            // device memory: the same granules as the launch's tail workgroup republishes them (the launch's one decision)
    unsigned long long *spec_status;        // pinned host memory: {2, tag} when the tail workgroup gave up waiting for the host
    int req_direct;                         // req_host is DEVICE memory the host writes through the PCIe BAR: the rollout workgroups poll it themselves (no republishing hop)
    uint32_t req_tag;                       // this launch's request tag (unique per prelaunch: a cancelled launch and its successor never share one)
    int req_polls;                          // how often the tail workgroup looks for the request before it gives up (~2 us a look)
    int no_early_mail;   // experiment switch (BN_NO_EARLY_MAIL): the self tail posts the first action behind its full merge, not from wave 0's light poll
    int aux_first;       // one-wave kernel, experiment: the aux workgroups (previous solve's tails) take the FIRST grid rows instead of the last
    int state_inline;    // the (one) instance's state travels in the kernel arguments (sv): a host loop that hands over a fresh state
    float sv0, sv1, sv2; // (three scalars, not an array: the tail workgroup works on a modified copy of this struct, and an array member
                         // of a copied struct lands in a scratch segment -- tests/test_build_artifacts.py)
                         // every control step (bn_mppi_forward_state_async) pays neither an upload nor the prologue's fetch of it
};

constexpr int kFlagStride = 32;            // spacing of the overlap counters, in counters of 8 bytes: 256 bytes, one memory channel each

// Launch grid of the rollout kernels.  x = workgroup-of-instance index interleaved with 2^xs instances, y = groups of
// 2^xs instances, then -- LAST in dispatch order -- the rows holding the B aux workgroups (tail of the previous solve):
// they start in the slots the first rollout workgroups free, instead of pushing the last instances' rollouts into a second
// residency round (64 x 16 rollout workgroups of K=1024 fill the chip's 4 x 256 slots exactly).
inline dim3 rollout_grid(const SolveParams &p, bool aux)
{
    const unsigned gx = (unsigned)p.nblk << p.xs;
    const unsigned rows = ((unsigned)p.B + (1u << p.xs) - 1) >> p.xs;
    const unsigned n_aux = (aux ? (unsigned)p.B : 0u) + (p.self_tail ? (unsigned)p.B : 0u);      // previous tails first, then the own ones
    const unsigned aux_rows = (n_aux + gx - 1) / gx;
    return dim3(gx, rows + aux_rows);
}

size_t rollout_lds_bytes(const SolveParams &p);
size_t finish_lds_bytes(const SolveParams &p);
size_t finish_lds_bytes_for(SolveParams p, bool sampled);   // ... before p.slip_on is set (bn_mppi_create)
size_t wave_lds_bytes(const SolveParams &p);
size_t lat_lds_bytes(const SolveParams &p);       // 0 when the latency variant cannot take this configuration
int rollout_blocks_per_cu(const SolveParams &p);   // runtime's occupancy answer for the headline rollout kernel (diagnostics)

hipError_t launch_rollout(const SolveParams &p, EpsMode mode, hipStream_t s);
hipError_t launch_rollout_lat_self(const SolveParams &p, EpsMode mode, hipStream_t s);   // the one-launch latency kernel (rollout_lat.inc mode 1: the solve's own tail in the launch)
hipError_t launch_rollout_lat_host(const SolveParams &p, hipStream_t s, hipEvent_t stop);       // the host-paced latency kernel (Philox noise), rollout_lat_host.hip
hipError_t launch_rollout_lat_host_ref(const SolveParams &p, hipStream_t s, hipEvent_t stop);   // ... in the reference's operation order
hipError_t launch_finish(const SolveParams &p, hipStream_t s);
// K-sharded solve: merge of all shards' partial rows -> ustar_cur, stats_cur, ustar, mean.  group_rows: 64 x (2 + 2T) floats, ticket: one zeroed int
hipError_t launch_shard_merge(const SolveParams &p, float *group_rows, int *ticket, hipStream_t s);
// rows idx[0..n) (nullptr: 0..n) of instance b's trajectory batch of the solve described by p (p.solve, p.eps, p.state, p.mean_used)
hipError_t launch_reroll(const SolveParams &p, EpsMode mode, int b, const int *idx, int n, float *out_n_T1_3, hipStream_t s);
hipError_t launch_rollout_sampled(const SolveParams &p, EpsMode mode, hipStream_t s);
size_t sampled_resident_per_cu(const SolveParams &p);   // workgroups of that kernel per CU (LDS, wave slots)
bool sampled_fused(const SolveParams &p);   // the sampled launch merges by ticket and carries the previous tail (LDS-window variant)
hipError_t launch_dwa(const SolveParams &p, const float *actions, const float *stage_goal, int NA, float *Xall, float *cost,
                      float *w, int *best, float *best_states, float *best_action, hipStream_t s);
// DWA's host geometry on the device: window grid (B, nv*nw, 2) around prev_action (B,2) and the sub-goal (B,2) on `path` (P,2)
hipError_t launch_dwa_window(const SolveParams &p, const float *prev_action, const float a_lim[2], float dwa_dt, int nv, int nw,
                             const float *path, int P, float lookahead, float *actions, float *stage_goal, hipStream_t s);

hipError_t launch_env_step(const SolveParams &p, const float *actions, float *states, float *reward, int *terminated, const float *z,
                           uint64_t step, hipStream_t s);
hipError_t launch_env_collision(const SolveParams &p, const float *states, int N, float thr, const float *z, uint64_t draw,
                                unsigned char *out, hipStream_t s);

// flag_and_seen: two zeroed device ints.  A grid on `waiter` that the chip cannot hold at once stays until a kernel on `setter` has
// set the flag (bounded) and records in [1] whether its first workgroup saw it: only if the two streams dispatch concurrently.
hipError_t launch_queue_probe(int *flag_and_seen, hipStream_t waiter, hipStream_t setter, int n_cus);
// every float d in [0, d_max]: does quotient_general's fast form give floor(d / res) (and the same bits for d >= 1e-30)?  *bad = count of failures
hipError_t launch_echo64(const unsigned long long *src, unsigned long long *dst, hipStream_t s);   // one thread: *dst = *src (system scope both ways)
hipError_t launch_stamp(int *word, int value, hipStream_t s);      // one thread: *word = value (system scope) -- a marker in a queue, for the host
hipError_t launch_quotient_check(float res, float inv_res, float d_max, unsigned long long *bad, hipStream_t s);
hipError_t launch_math_eval(int fn, const float *in, float *out, size_t n, hipStream_t s);   // 0 sqrt, 1 sin, 2 cos, 3 wrap, 4 wrap_near

// layout conversion helpers (planner-native k-fastest <-> reference k-major)
hipError_t launch_states_to_reference(const float *X_soa, float *X_aos, int K, int Kp, int T1, hipStream_t s);   // (T1,3,Kp)->(K,T1,3)
hipError_t launch_controls_to_reference(const float *U_soa, float *U_aos, int K, int Kp, int T, hipStream_t s);  // (T,2,Kp)->(K,T,2)
hipError_t launch_gather_states(const float *X_soa, const int *idx, float *out, int n, int Kp, int T1, hipStream_t s);
hipError_t launch_philox_slip(float *zt, float *zc, float *zo, uint64_t seed, uint64_t solve, int b, int K, int T, hipStream_t s);   // (K,T), (K,T+1), (T)
hipError_t launch_philox_noise(float *eps_kt2, uint64_t seed, uint64_t solve, int b, int K, int T, int k0, hipStream_t s);

}  // namespace bn
