/*
 * benchnav_mppi.h -- C ABI of the MI355X-native MPPI local planner.
 *
 * This is the drop-in boundary for BenchNav's MPPI hot path.  The reference has
 * no FFI layer: its boundary is the Python class
 *     src/planners/local_planners/mppi.py:17   class MPPI(nn.Module)
 * whose methods the entry points below replace one for one (citations are
 * reference file:line).  The Python mirror of that class lives in
 * benchnav_amd/mppi.py and binds these symbols with ctypes; INTEGRATION.md shows
 * the stub a BenchNav maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns BN_OK (0) or a negative bn_status,
 *     bn_last_error() gives the message of the calling thread's last failure;
 *   - a handle is bound to one HIP device and one HIP stream; it is not
 *     thread-safe, distinct handles are independent;
 *   - "host"/"device" in a parameter name says where the pointer must live;
 *     the library owns every device buffer it allocates, callers own theirs;
 *   - all arrays are float32, C-contiguous; shapes use the reference's names:
 *     K = num_samples, T = horizon, G = grid_size, B = num_instances;
 *   - `*_async` entry points only enqueue work on the handle's stream.
 */
#ifndef BENCHNAV_MPPI_H
#define BENCHNAV_MPPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN_MPPI_ABI_VERSION 4

typedef enum bn_status {
    BN_OK = 0,
    BN_ERR_INVALID = -1,      /* bad argument / unsupported configuration          */
    BN_ERR_HIP = -2,          /* a HIP runtime call failed (message has the detail) */
    BN_ERR_NO_DEVICE = -3,    /* no usable gfx950 device                            */
    BN_ERR_STATE = -4         /* call sequence error (e.g. solve before set_map)    */
} bn_status;

/* Where the per-solve standard-normal noise eps comes from (mppi.py:149-151). */
typedef enum bn_noise_kind {
    BN_NOISE_PHILOX = 0,      /* generated in the rollout kernel (Philox4x32 + Box-Muller: eight rounds for the per-rollout streams) */
    BN_NOISE_HOST_KT2 = 1,    /* host pointer, (B,K,T,2): the reference's rsample layout       */
    BN_NOISE_DEVICE_KT2 = 2,  /* device pointer, (B,K,T,2)                                      */
    BN_NOISE_DEVICE_T2K = 3   /* device pointer, (B,T,2,K): planner-native, coalesced           */
} bn_noise_kind;

typedef enum bn_mem_kind { BN_MEM_HOST = 0, BN_MEM_DEVICE = 1 } bn_mem_kind;

/* Device buffers a caller may map without a copy (bn_mppi_device_buffer). */
typedef enum bn_buffer_id {
    BN_BUF_STATES = 0,    /* (B,T+1,3,Kp) _state_seq_batch, planner-native layout: k fastest, rows pitched to
                             Kp = 64*ceil(K/64) floats (bn_mppi_row_pitch)                          */
    BN_BUF_WEIGHTS = 1,   /* (B,K)        _weights                                             */
    BN_BUF_COSTS = 2,     /* (B,K)        per-rollout total cost                               */
    BN_BUF_CONTROLS = 3,  /* (B,T,2,Kp)   _perturbed_action_seqs (only with BN_FLAG_STORE_CONTROLS) */
    BN_BUF_USTAR = 4,     /* (B,T,2)      optimal_action_seq                                   */
    BN_BUF_XSTAR = 5,     /* (B,T+1,3)    optimal_state_seq                                    */
    BN_BUF_MEAN = 6,      /* (B,T,2)      _previous_action_seq                                 */
    BN_BUF_MAP = 7,       /* (n_maps,G,G) risk map(s)                                          */
    BN_BUF_GOAL = 8,      /* (B,2)                                                             */
    BN_BUF_USTAR_XSTAR = 9, /* the (B,T,2) U* block followed by the (B,T+1,3) X* block: both outputs of forward() are one
                               contiguous allocation, so a caller that wants private copies makes ONE device copy     */
    BN_BUF_STATES_ALT = 10,   /* the second (B,T+1,3,Kp) / (B,T,2,Kp) buffers of handles that overlap launches (bn_mppi_states_buffer_index) */
    BN_BUF_CONTROLS_ALT = 11,
    BN_BUF_COUNT_ = 12
} bn_buffer_id;

enum {
    BN_FLAG_STORE_CONTROLS = 1u << 0,  /* materialise _perturbed_action_seqs in HBM          */
    BN_FLAG_SHARED_MAP = 1u << 1,      /* all B instances plan on map 0                      */
    BN_FLAG_NO_LDS_WINDOW = 1u << 2,   /* debug: gather the risk map from global memory      */
    BN_FLAG_PROFILE = 1u << 3,         /* record HIP events around each kernel (see bn_mppi_kernel_ms) */
    BN_FLAG_PRIVATE_STREAM = 1u << 4,  /* ignore `stream`: the library creates and owns a non-blocking stream */
    BN_FLAG_NO_PIPELINE = 1u << 5,     /* always two launches per solve (rollout, finish); see bn_mppi_solve_async */
    BN_FLAG_WAVE_KERNEL = 1u << 7,     /* always use the one-wave-per-64-rollouts throughput kernel (default: chosen by launch size) */
    BN_FLAG_ROLE_KERNEL = 1u << 8,     /* always use the five-wave role-split latency kernel */
    BN_FLAG_NO_OVERLAP = 1u << 11,     /* bn_mppi_solve_n_async keeps all its launches on the handle's stream (default: with the latency
                                          kernel consecutive solves alternate between two streams and overlap, see there) */
    BN_FLAG_LAT_KERNEL = 1u << 10,     /* always use the barrier-free latency variant of the role kernel when it fits (default: when the
                                          launch leaves every workgroup a CU to itself) */
    BN_FLAG_LEAN = 1u << 9,            /* lean mode: _state_seq_batch (mppi.py:119-125) is not materialised -- 70 % of a solve's
                                          HBM bytes; bn_mppi_get_states / bn_mppi_get_top_samples / bn_mppi_reroll_async
                                          regenerate the requested rows of the latest solve bit-identically on demand */
    BN_FLAG_SAMPLED_SLIP = 1u << 6,    /* BASELINE config 3: every traversability lookup of the rollouts draws
                                          slip ~ Normal(map, slip_std)[cell]; see bn_mppi_set_slip_std */
    BN_FLAG_HOST_PACED = 1u << 13,     /* opt-in for a host loop that calls bn_mppi_forward_state_async once per control step and consumes every solve's
                                          first action (test_mppi.py:174-181): the launch of the NEXT solve is enqueued one step ahead and waits on
                                          the device for the state the next call posts -- see bn_mppi_forward_state_async.  One instance, latency
                                          kernel, num_samples <= 1024, Philox noise; bn_mppi_host_paced() tells whether the handle qualifies */
    BN_FLAG_UNORDERED_OUTPUTS = 1u << 14, /* with BN_FLAG_HOST_PACED, for a loop that consumes the first action and rarely anything else: the steady-state
                                          bn_mppi_forward_state_async does NOT order the handle's stream behind the posted solve (3-5 us of host
                                          time per control step).  Every other entry point of the handle (bn_mppi_first_action and the pure queries
                                          aside) makes up for it when it is called; a caller that enqueues its OWN consumers of `out_device` or of
                                          the device buffers calls bn_mppi_order_outputs first */
    BN_FLAG_REFERENCE_ORDER = 1u << 12 /* the transit in the REFERENCE's own operation order, robot_model.py:75-95: sin / cos of every
                                          step's heading, x + ((trav v) cos) dt, theta + (trav omega) dt, the general heading wrap.
                                          The default arithmetic (heading vector carried by a rotation per step, one fused update) leaves
                                          positions 1-2 ulp from the reference's, and about one rollout in 30 000 (T = 50; one in 7 000 at
                                          T = 100) then lands in a neighbouring cell and leaves the 1e-4 trajectory tolerance from there on
                                          (tests/golden/census_*.npz); with this flag none did in 0.7 M rollouts at T = 50 and
                                          one in 0.7 M at T = 100 -- the level of libm against the reference's own SLEEF.  Every kernel has
                                          an instantiation in this arithmetic; the price is the chain's extra instructions: 10.8 instead of
                                          8.0 us per dependent single-instance solve, 1-8 % for batched launches and K > 4096 (DESIGN.md
                                          4.15).  dt * max|omega| > 0.5 selects this arithmetic
                                          by itself).  bn_mppi_arithmetic() tells which arithmetic a handle runs. */
};

/*
 * Everything MPPI.__init__ (mppi.py:23-128) receives or reads from its
 * `dynamics` / `objectives` arguments, flattened to plain data:
 *   horizon, num_samples              mppi.py:25-26
 *   sigma                             mppi.py:31,89
 *   inv_var = diag(inverse(diag(sigma^2)))   mppi.py:94-97 (host computes it the reference's way)
 *   lambda_                           mppi.py:32,90
 *   u_min, u_max                      robot_model.py:54-57 via mppi.py:83-88
 *   dt                                robot_model.py:60 (transit's default delta_t = 0.1)
 *   stuck_threshold                   objectives.py:27
 *   grid_size, resolution, x/y_limits grid_map.py:40-50  (limits[0] is also the index origin, grid_map.py:199-201)
 */
typedef struct bn_mppi_config {
    uint32_t struct_size;     /* = sizeof(bn_mppi_config) */
    int32_t device_id;        /* HIP device ordinal */
    int32_t horizon;          /* T >= 1 */
    int32_t num_samples;      /* K >= 1 */
    int32_t num_instances;    /* B >= 1 independent planning instances solved per call */
    int32_t grid_size;        /* G >= 1 */
    float resolution;
    float x_limits[2];
    float y_limits[2];
    float sigma[2];
    float inv_var[2];
    float lambda_;
    float u_min[2];
    float u_max[2];
    float dt;
    float stuck_threshold;
    uint64_t seed;            /* Philox key for BN_NOISE_PHILOX */
    uint32_t flags;           /* BN_FLAG_* */
    void *stream;             /* hipStream_t to enqueue on; NULL is HIP's default (null) stream, which is
                                 also torch's default stream.  See BN_FLAG_PRIVATE_STREAM. */
} bn_mppi_config;

typedef struct bn_mppi bn_mppi_t;

/* Fills *cfg with the reference's defaults (robot_model.py:54-60, test_mppi.py:160-169). */
void bn_mppi_config_init(bn_mppi_config *cfg);

/* MPPI.__init__, mppi.py:23-128: allocates the device buffers (_state_seq_batch,
 * _weights, _previous_action_seq = 0, ...).  Fails with BN_ERR_NO_DEVICE when no
 * gfx950 device is present: there is no CPU fallback. */
int bn_mppi_create(const bn_mppi_config *cfg, bn_mppi_t **out);
void bn_mppi_destroy(bn_mppi_t *h);

/* dynamics._traversability_model._risks (traversability_model.py:24-26), (G,G)
 * row-major [iy][ix] (grid_map.py:167).  instance = -1 sets every map. */
int bn_mppi_set_map(bn_mppi_t *h, int32_t instance, const float *risk, bn_mem_kind where);
/* Sampled-slip mode (BN_FLAG_SAMPLED_SLIP): the map set with bn_mppi_set_map is the slip MEAN and this the slip
 * STD per cell, (G,G); every get_traversability of the rollouts then evaluates 1 - clamp(z*std + mean, 0, 1) with
 * a fresh standard normal z (the observation-mode branch, traversability_model.py:65-69): T draws in transit,
 * T+1 in the stage/terminal costs per rollout, T in the optimal rollout.  The reference's own MPPI cannot run in
 * this mode; the semantics are those of its components. */
int bn_mppi_set_slip_std(bn_mppi_t *h, int32_t instance, const float *std, bn_mem_kind where);
/* Optional injected draws for the next solves (device pointers, k fastest): transit (B,T,K), cost (B,T+1,K),
 * optimal rollout (B,T).  NULL (default) = the library's Philox stream. */
int bn_mppi_set_slip_noise(bn_mppi_t *h, const float *zt_device, const float *zc_device, const float *zo_device);
/* objectives._goal_pos (objectives.py:26).  instance = -1 sets every instance. */
int bn_mppi_set_goal(bn_mppi_t *h, int32_t instance, const float goal_host[2]);
/* _previous_action_seq (mppi.py:116,217): (T,2) host array; NULL resets it to zero. */
int bn_mppi_set_mean(bn_mppi_t *h, int32_t instance, const float *mean_host);
int bn_mppi_get_mean(bn_mppi_t *h, int32_t instance, float *mean_host);

/*
 * MPPI.forward(state), mppi.py:130-219, for all B instances.
 *   states   (B,3) current states, host or device per `states_where`
 *   eps      noise per `noise` (NULL for BN_NOISE_PHILOX)
 *   ustar_host  (B,T,2)    optimal_action_seq, may be NULL
 *   xstar_host  (B,T+1,3)  optimal_state_seq,  may be NULL
 * Synchronous like the reference: returns after the stream has drained and the
 * outputs are in the host arrays.  Updates _previous_action_seq (no shift, mppi.py:217).
 */
int bn_mppi_solve(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps,
                  bn_noise_kind noise, float *ustar_host, float *xstar_host);
/* Same, enqueue only: results stay in the device buffers (bn_mppi_device_buffer).
 * For K <= 2048 consecutive async solves are software-pipelined: the launch of solve i also merges
 * solve i-1's softmin statistics (its warm start) and writes solve i-1's U*, X* and weights, so a
 * dependent chain of solves costs one launch each.  bn_mppi_sync, bn_mppi_solve and every getter
 * first write the pending tail of the latest solve; device buffers obtained with
 * bn_mppi_device_buffer are up to date after bn_mppi_sync (or bn_mppi_flush + stream order). */
int bn_mppi_solve_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where,
                        const float *eps, bn_noise_kind noise);
/* MPPI.forward (mppi.py:130-219) for a host loop that consumes every solve (test_mppi.py:174-183): bn_mppi_solve_async +
 * bn_mppi_flush in one call.  states_device (B,3) and eps_device are device pointers; nothing waits.  out_device (may be
 * NULL): a caller-owned device block laid out like BN_BUF_USTAR_XSTAR that the tail fills as well -- the fresh
 * (optimal_action_seq, optimal_state_seq) tensors the reference returns, without another launch. */
int bn_mppi_forward_async(bn_mppi_t *h, const float *states_device, const float *eps_device, bn_noise_kind noise,
                          float *out_device);
/* The same with the states taken from HOST memory, by value, at the call (ABI 4): the loop of test_mppi.py:174-181 whose environment
 * lives on the host has a fresh state every control step, and `state.to(device)` in front of forward() (mppi.py:140-144) is an
 * upload the solve then waits for.  states_host (B,3) may be reused as soon as the call returns.  One instance on the latency
 * kernel: the state travels in the kernel arguments -- no copy is enqueued, and the kernel's prologue has no fetch in front of its
 * window; other configurations stage and upload it (as bn_mppi_solve_async does for BN_MEM_HOST).
 * ONE LAUNCH per call (both entry points, and bn_mppi_solve) where the latency kernel serves the handle (bn_mppi_launches_per_forward):
 * the tail of the solve -- softmin merge, U*, the first-action mailbox (bn_mppi_first_action), X*, the weights -- is a workgroup of
 * the rollout launch itself that waits on the device for the rollout workgroups dispatched in front of it; results are bit-identical
 * to the two-launch path's.  That wait is bounded like every device-side wait of the library and cannot expire short of a stalled
 * GPU; if it does, the next call that concerns results returns BN_ERR_HIP (the latest solve's outputs are invalid) and the handle
 * keeps to two launches from then on. */
int bn_mppi_forward_state_async(bn_mppi_t *h, const float *states_host, const float *eps_device, bn_noise_kind noise,
                                float *out_device);
/* With BN_FLAG_HOST_PACED (and BN_NOISE_PHILOX) the call also enqueues the launch of the NEXT solve -- on a stream of the library's
 * own -- before it returns: while the host waits for this solve's first action, takes its environment step and comes back, that
 * launch gets its latency, its instruction fetch, the horizon's noise, its mean (the merge of this solve's partial rows), the whole
 * control tile and a window around THIS state (two cells wider each side) behind it and then waits, on the device, for the state the
 * next call posts to pinned host memory.  Between the host's store and the first action's way back only the rollouts' 50-step chain,
 * the cost epilogue and the merge are left: 25 -> ~14 us per control step at K = 1024, T = 50.  Results are bit-identical to the
 * one-launch path's (same noise stream positions, same merges).
 *   - outputs stay stream-ordered for what comes AFTER: every posted solve's launch is followed by an event the handle's stream waits
 *     for (BN_FLAG_UNORDERED_OUTPUTS: not before the next call that needs it, see bn_mppi_order_outputs).  What was enqueued on the handle's stream BEFORE the call is not waited for (the launch is paced by the host's store, not by
 *     the queue): work left there that still reads the planner's own buffers (BN_BUF_WEIGHTS, BN_BUF_STATES, BN_BUF_USTAR_XSTAR ...),
 *     or that still uses the memory `out_device` was carved from, must have completed -- fresh output blocks per call need nothing;
 *   - a launch that waits for a state holds 17 CUs and is cancelled -- a word in pinned memory, no synchronisation -- by every other
 *     entry point of the handle except bn_mppi_first_action and the pure queries (the next bn_mppi_forward_state_async then pays one
 *     ordinary launch); left alone it gives up after ~50 ms and the next call starts over the same way;
 *   - a DEVICE-WIDE synchronisation (hipDeviceSynchronize, torch.cuda.synchronize) issued while a launch waits blocks for up to
 *     those 50 ms: end the loop with any other call of the handle (bn_mppi_flush is the cheapest) first.  This is why the mode is
 *     opt-in;
 *   - a state more than two cells from the previous one (a reset) is handled inside the waiting launch (it stages its window again). */
/* BN_FLAG_UNORDERED_OUTPUTS: order the handle's stream behind the latest posted solve (one stream-wait; a no-op when that has been done
 * or the flag is not set).  Does not cancel the launch that waits for the next state: the loop keeps its pace. */
int bn_mppi_order_outputs(bn_mppi_t *h);
int32_t bn_mppi_host_paced(const bn_mppi_t *h);      /* 0: no; 1: BN_FLAG_HOST_PACED was given and the handle qualifies; 2: ... and the request words live in
                                                         device memory the host writes through the PCIe BAR (no PCIe read on the launch's side) */
/* Which of the two trajectory / control buffers holds the LATEST solve (0: BN_BUF_STATES / BN_BUF_CONTROLS, 1: the *_ALT ones): host-paced
 * solves alternate between them (two launches are in flight and must not write the same addresses); every other path writes buffer 0. */
int32_t bn_mppi_states_buffer_index(const bn_mppi_t *h);
/* n dependent solves enqueued from one call (the warm start chains them on the device; the state is
 * re-read from `states` by every solve).  Noise block i is eps + (i % eps_ring) * eps_stride floats.
 * With n >= 3 (and device-resident inputs) the solves of one call alternate between the handle's stream and an internal one and
 * OVERLAP: solve i+1 is dispatched while solve i runs and waits on the device for its predecessor's softmin partials instead of
 * for its kernel's end; results are bit-identical to the one-stream chain (BN_FLAG_NO_OVERLAP turns it off: the switch for a GPU
 * shared with other work).  Several handles in one process: a batch overlaps only while every other handle on the device is idle,
 * and a launch of another handle that arrives while an overlapped batch is in flight runs behind that batch's end (events on the
 * device, no host blocking) -- waiting workgroups hold their slots and must not share the device with anybody's launches.
 * What is ordered for the caller:
 *   - the LAST solve of the call and the tail behind it run on the handle's stream, so work enqueued there afterwards is ordered
 *     behind the batch's final results by the queue itself; the internal stream is not joined back -- its kernels have handed
 *     everything over through device counters by then and end within microseconds (bn_mppi_sync waits for both);
 *   - `states` is read by every solve of the call, `eps` block i by solve i: they must stay valid and unchanged until the batch has
 *     run -- work enqueued on the handle's stream behind the call may overwrite `states` (a host-driven closed loop that keeps
 *     ONE state buffer: batch, write the next state in stream order, batch);
 *   - `eps` (caller-provided noise) must stay valid AND UNCHANGED until the synchronisation point that follows the batch
 *     (bn_mppi_sync, a getter, or the caller's own stream synchronisation): a re-run, see below, reads it again.
 * Device-side waits are bounded (~2 s: another user of the GPU kept a predecessor from becoming resident).  If one expires the
 * launch computes on incomplete partials; the error word it sets lives in pinned host memory and is looked at by EVERY entry
 * point that synchronises (bn_mppi_sync, the getters, the setters, bn_mppi_episode_log ...) and by bn_mppi_flush: the batches
 * enqueued since the last clean synchronisation point are then RE-RUN on one stream from the mean the first of them started
 * from (kept on the device), with the same Philox positions, the callers' noise buffers, and the states each batch was GIVEN: the
 * first launch of every journalled batch keeps them in memory the handle owns (ABI 3, round 4; before that a caller that rewrote
 * its state tensor in place between batches got the re-run on the new contents).
 * The call returns BN_OK with a warning in bn_last_error(), bn_mppi_recovery_count() counts these events (consumers enqueued in
 * stream order BEFORE the synchronisation point have read invalid buffers), and the handle keeps to one stream from then on.
 * BN_ERR_HIP only when the batches cannot be re-run (host-resident inputs; more than min(4096, 16 MB / (12 B x num_instances), at least 64) batches without a synchronising call).
 * Host side of the call.  "async" means the call does not wait for the solves -- but it is not free of host waits: a batch whose
 * launches exceed one residency round each (64+ instances of K = 1024 on the role kernel) hands its first three launches over one
 * by one -- the host spins on a pinned word (a few microseconds; at most 2 ms per hand-over, then it falls back to
 * hipStreamSynchronize on that launch's stream, which also waits for whatever the caller queued there before) --, and a batch of
 * such launches that finds the handle's stream busy synchronises it first.  Small launches (the single-instance path) never block.
 * Threads.  A handle is not thread-safe, and the handles of ONE device are not independent of each other while overlapped batches
 * are in use: the library orders every launch of a handle behind another handle's overlapped batch that may still be in flight
 * (solves, stand-alone tails, re-rolls, environment steps), and it decides that from host-side state it reads at the call.  Calls
 * that concern one device must therefore come from one thread at a time (the caller's lock, or one thread per device); handles on
 * different devices are independent.  BN_FLAG_NO_OVERLAP on every handle of a device lifts the restriction for those handles. */
int bn_mppi_solve_n_async(bn_mppi_t *h, int32_t n, const float *states, bn_mem_kind states_where,
                          const float *eps, bn_noise_kind noise, int32_t eps_ring, int64_t eps_stride);

/* ---- K-sharded solve (optional row): ONE solve whose K_total rollouts are split over ranks ----------
 * Every rank owns a handle with num_samples = its shard and rollouts [first_rollout, first_rollout + num_samples) of
 * the global solve (the offset keys the Philox stream, so the union of the shards draws exactly the noise of the
 * unsharded solve).  Per solve:
 *   1. bn_mppi_shard_rollout_async            rollouts + costs of the shard, per-workgroup softmin partials
 *   2. exchange                               all-gather bn_mppi_shard_partials over the ranks, rank order (RCCL)
 *   3. bn_mppi_shard_finish_async             merge ALL partials in that order (bit-identical U* on every rank and equal
 *                                             to the unsharded solve's), next mean, X*, and the shard's weights
 * The reference has no multi-GPU path; the exchange is the log-sum-exp merge of mppi.py:193-199. */
int bn_mppi_set_rollout_offset(bn_mppi_t *h, int64_t first_rollout);
int bn_mppi_shard_rollout_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps, bn_noise_kind noise);
int bn_mppi_shard_partials(bn_mppi_t *h, const float **partials_device, int32_t *workgroups, int32_t *floats_per_workgroup);
int bn_mppi_shard_finish_async(bn_mppi_t *h, const float *all_partials_device, int32_t total_workgroups);
/* The same three steps as ONE call with the exchange enqueued by the library: RCCL's ncclAllGather on the handle's own stream
 * between the two kernels -- no host wait, no event, no second stream (what a caller driving steps 1-3 through
 * torch.distributed pays per solve).  Equal shards only (num_samples the same on every rank).  RCCL is opened with dlopen
 * on first use; nothing else in the library depends on it.
 *   bn_dist_unique_id          rank 0 draws the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means
 *                              (benchnav_amd.sharding broadcasts them over the torch.distributed group the job has anyway)
 *   bn_mppi_shard_comm_init    collective over the ranks of the solve (ncclCommInitRank on the handle's device); the communicator
 *                              and the buffer of gathered partial rows belong to the handle and go with bn_mppi_destroy
 *   bn_mppi_shard_solve_async  rollouts of the shard, all-gather of the partial rows (rank order), merge + tail
 * Results are those of steps 1-3 above: bit-identical on every rank and to the unsharded solve.  Ordering: U* and the next mean are
 * written on the handle's stream by the merge; X*, the shard's weights and the cost copy by a tail the library runs on a second
 * stream beside the NEXT solve's rollouts -- they are ordered on the handle's stream by the next call that concerns results
 * (bn_mppi_flush, bn_mppi_sync, any getter, bn_mppi_device_buffer), which enqueues the join without blocking the host. */
#define BN_DIST_UNIQUE_ID_BYTES 128
int bn_dist_unique_id(uint8_t out[BN_DIST_UNIQUE_ID_BYTES]);
/* Everything bn_mppi_shard_comm_init can fail on short of the collective itself, done locally (ABI 4): RCCL opened, the handle's state
 * checked, the gathered-rows buffer, the events and the side stream created.  A job calls it on every rank, agrees on the outcome
 * (one all-reduce over the process group it has anyway) and enters bn_mppi_shard_comm_init -- ncclCommInitRank, which blocks until
 * EVERY rank has entered -- only if all of them are ready.  bn_mppi_shard_comm_init calls it itself when the caller has not. */
int bn_mppi_shard_comm_prepare(bn_mppi_t *h, int32_t world_size, int32_t rank);
int bn_mppi_shard_comm_init(bn_mppi_t *h, const uint8_t unique_id[BN_DIST_UNIQUE_ID_BYTES], int32_t world_size, int32_t rank);
int bn_mppi_shard_solve_async(bn_mppi_t *h, const float *states, bn_mem_kind states_where, const float *eps, bn_noise_kind noise);
/*
 * Device-side closed loop: PlanetaryEnv.step (planetary_env.py:189-219) between consecutive solves, for
 * all B instances, without a host round trip per control step (the reference loop: test_mppi.py:171-198).
 *   env_attach   latent slip model Normal(mean, std) per cell, (n_maps,G,G) each (grid_map.distributions
 *                ["latent_models"], sampled in observation mode: traversability_model.py:65-69),
 *                goal_threshold (planetary_env.py:215-217), the environment's delta_t, Philox key.
 *   episode      n_steps control steps: solve at the current state, apply U*[0] through the observation-mode
 *                transit with a freshly sampled slip, test the goal.  Like PlanetaryEnv.step, a terminated
 *                environment keeps moving when stepped (the reference's driver loop stops instead,
 *                test_mppi.py:192-194); bn_mppi_env_set_freeze(h, 1) opts into freezing an instance once it is
 *                within goal_threshold, for fixed-length batched episodes.  z_device: optional (n_steps, B) standard normals for the slip draws (parity
 *                tests), NULL = in-kernel Philox.  One launch per control step (K <= 2048 only).
 *   episode_log  states (n_steps+1, B, 3) incl. the initial one, rewards (n_steps, B) = traversability
 *                observed by each step (planetary_env.py:209), actions (n_steps, B, 2) = the control each step
 *                applied (action_seq[0] of test_mppi.py:181), done_step (B) = first step whose resulting
 *                state lies within goal_threshold, or -1.  Any pointer may be NULL.  Synchronises.
 */
int bn_mppi_env_attach(bn_mppi_t *h, const float *latent_mean, const float *latent_std, bn_mem_kind where,
                       float goal_threshold, float delta_t, uint64_t seed);
/* The two PlanetaryEnv calls on their own, for callers that drive the loop themselves (all pointers device memory, stream-ordered):
 *   env_step             planetary_env.py:189-219 for the B environments: states (B,3) advance in place through the
 *                        observation-mode transit with actions (B,2); rewards (B,) = the sampled traversability;
 *                        terminated (B,) = within goal_threshold of the goal (a terminated environment moves on when
 *                        stepped again, as the reference's does, unless bn_mppi_env_set_freeze).  z (B,) injects the slip draws, NULL = Philox keyed by
 *                        (env seed, step_index).
 *   env_collision_check  planetary_env.py:221-232: out (B,N) = sampled traversability at states (B,N,3) <= stuck_threshold,
 *                        one fresh draw per position (z (B,N) injects them; NULL = Philox keyed by (env seed, draw_index)). */
int bn_mppi_env_set_freeze(bn_mppi_t *h, int32_t freeze_on_goal);      /* default 0 = the reference's behaviour */
int bn_mppi_env_step(bn_mppi_t *h, const float *actions_device, float *states_device, float *rewards_device,
                     int32_t *terminated_device, const float *z_device, uint64_t step_index);
int bn_mppi_env_collision_check(bn_mppi_t *h, const float *states_device, int32_t n_positions, float stuck_threshold,
                                const float *z_device, uint64_t draw_index, uint8_t *out_device);
int bn_mppi_episode_async(bn_mppi_t *h, int32_t n_steps, const float *states0, bn_mem_kind states_where,
                          const float *eps, bn_noise_kind noise, int32_t eps_ring, int64_t eps_stride,
                          const float *z_device);
int bn_mppi_episode_log(bn_mppi_t *h, float *states_host, float *rewards_host, float *actions_host,
                        int32_t *done_step_host);
/*
 * DWA.forward(state), reference src/planners/local_planners/dwa.py:116-153, on the planner handle's map, goal and
 * geometry.  The caller supplies the dynamic-window candidates (dwa.py:168-199: cartesian_prod of two linspaces),
 * (B, num_actions, 2) constant (v, omega) pairs; each is rolled out `horizon` steps with the MPPI transit/aliasing
 * (dwa.py:224-227), costed against `stage_goal` (the sub-goal of dwa.py:260-285, NULL = the goal) per stage and
 * against the goal at the end, summed in fp32 in step order (dwa.py:251-256).
 *   best_action (B,2), best_states (B,T+1,3): the argmin candidate (first minimum)        dwa.py:139-143
 *   costs, weights (B,num_actions): cost_batch and softmax(-cost_batch)                    dwa.py:151
 *   states_all (B,num_actions,T+1,3): _state_seq_batch; best_index (B).  Any output may be NULL.  Synchronous.
 */
int bn_mppi_dwa_solve(bn_mppi_t *h, const float *states_host, const float *actions_host, int32_t num_actions,
                      const float *stage_goal_host, float *best_action_host, float *best_states_host,
                      float *costs_host, float *weights_host, float *states_all_host, int32_t *best_index_host);
/* Device-resident outputs of the latest bn_mppi_dwa_solve with this num_actions (valid until the next call that uses the
 * handle's scratch memory): states_all (B,num_actions,T+1,3), costs and weights (B,num_actions). */
int bn_mppi_dwa_buffers(bn_mppi_t *h, int32_t num_actions, const float **states_all_device, const float **costs_device,
                        const float **weights_device);
/* DWA.forward with nothing on the host (dwa.py:116-153 incl. _generate_actions :168-199 and the sub-goal of :240-244,
 * 260-285): the window grid around prev_action_device (B,2) -- linspace x linspace, v major -- and the sub-goal on
 * path_device (num_path,2; NULL: the goal) are computed on the device, the sub-goal like the reference from candidate 0's
 * aliased slot-0 state; then the candidates are rolled out and costed as in bn_mppi_dwa_solve.  Only enqueues.
 * prev_action_device is updated in place with the argmin action (dwa.py:147: the next window's centre) -- it IS
 * optimal_action_seq; best_states_device (B,T+1,3) may be NULL.  The candidate batch stays in the handle's scratch:
 * bn_mppi_dwa_buffers (states, costs, weights) and bn_mppi_dwa_candidates (the window grid and the sub-goal used). */
int bn_mppi_dwa_forward_async(bn_mppi_t *h, const float *states_device, float *prev_action_device, const float a_lim_host[2],
                              float dwa_delta_t, int32_t num_lin_vel, int32_t num_ang_vel, const float *path_device, int32_t num_path,
                              float lookahead, float *best_states_device);
int bn_mppi_dwa_candidates(bn_mppi_t *h, int32_t num_actions, const float **actions_device, const float **stage_goal_device);
/* Write the pending tail, wait for the handle's stream, check the overlapped launches' error word (see bn_mppi_solve_n_async). */
int bn_mppi_sync(bn_mppi_t *h);
/* optimal_action_seq[0] (mppi.py:219 -> test_mppi.py:181: what env.step consumes) of the latest solve, on the host, as early as it
 * exists: the tail of a solve posts U*[0] to pinned host memory right after the softmin merge -- before it rolls out X* and
 * normalises the weights -- and this call polls for it.  No stream synchronisation, no device-to-host copy; the other outputs are
 * complete in stream order as always.  Enqueues the pending tail first (like every getter). */
int bn_mppi_first_action(bn_mppi_t *h, int32_t instance, float action_host[2]);
/* How many times this handle re-ran batches after an expired device-side wait (0 in normal operation). */
uint64_t bn_mppi_recovery_count(const bn_mppi_t *h);
/* Which way bn_mppi_solve_n_async runs its batches (ABI 4).  Overlapped launches assume the DEVICE TO THEMSELVES: the workgroups of
 * launch i+1 wait, resident, for launch i's partials, and with another process's kernels on the GPU those waiting workgroups hold the
 * slots launch i's stragglers need -- the chain then runs at half the one-stream rate (and, in the extreme, a bounded wait expires:
 * see above).  The handle protects itself: it measures its own cadence over windows of 64 launches (the first-action mailbox counts
 * finished solves; no event, no kernel change), looks at the other mode once, keeps the faster one, looks again when its cadence
 * degrades by half and -- while on one stream -- every 256 windows.
 *   0  overlapped (two streams)      1  one stream, by the handle's own choice (somebody else is on the device)
 *   2  one stream for good: a device-side wait expired once (bn_mppi_recovery_count)      3  the handle never overlaps (flags, kernel family) */
int32_t bn_mppi_overlap_mode(const bn_mppi_t *h);
/* Test hook: feed the tuner a cadence (microseconds per launch) measured in `mode` (0 overlapped, 1 one stream); returns 1 when the
 * next long batch will look at the other mode. */
int bn_mppi_debug_cadence(bn_mppi_t *h, int32_t mode, double us_per_launch);
/* Test hook: behave as if a bounded device-side wait had expired in the batches enqueued since the last synchronisation point
 * (what happens when another process keeps a launch's predecessor from becoming resident for ~2 s): sets the error word and
 * overwrites what those batches wrote (mean, U* | X*, weights, costs, trajectories) with NaN patterns.  The next synchronising
 * call re-runs them on one stream. */
int bn_mppi_debug_expire_wait(bn_mppi_t *h);
/* Test hook (BN_FLAG_HOST_PACED): how many looks of ~2 us a prelaunched solve waits for its state before it gives up (default 25000), and
 * whether bn_mppi_forward_state_async posts the state without checking that the launch is still waiting. */
int bn_mppi_debug_host_paced(bn_mppi_t *h, int32_t polls, int32_t post_unchecked);
/* Enqueue the pending tail (if any) without waiting.  An expiry that has already been flagged is repaired here (that does wait). */
int bn_mppi_flush(bn_mppi_t *h);

/* Copies of the planner state in the REFERENCE's layouts (host arrays).  All
 * synchronise the stream first.
 *   weights  (K,)        _weights                 mppi.py:193
 *   costs    (K,)        `costs` local            mppi.py:186-190
 *   states   (K,T+1,3)   _state_seq_batch         mppi.py:119-125 (aliased slots)
 *   controls (K,T,2)     _perturbed_action_seqs   mppi.py:155-157 (needs BN_FLAG_STORE_CONTROLS)
 *   noise    (K,T,2)     eps of solve number `solve_index` regenerated from the Philox stream */
int bn_mppi_get_weights(bn_mppi_t *h, int32_t instance, float *out_host);
int bn_mppi_get_costs(bn_mppi_t *h, int32_t instance, float *out_host);
int bn_mppi_get_states(bn_mppi_t *h, int32_t instance, float *out_host);
int bn_mppi_get_controls(bn_mppi_t *h, int32_t instance, float *out_host);
int bn_mppi_get_philox_noise(bn_mppi_t *h, int32_t instance, uint64_t solve_index, float *out_host);
/* Sampled-slip mode: regenerate the library's slip draws of solve `solve_index` in the oracle's layout:
 * transit (K,T), cost (K,T+1), optimal rollout (T). */
int bn_mppi_get_slip_noise(bn_mppi_t *h, int32_t instance, uint64_t solve_index, float *zt_host, float *zc_host, float *zo_host);

/* MPPI.get_top_samples(n), mppi.py:221-240: the n highest-weight rollouts, sorted
 * by weight descending.  states_host (n,T+1,3), weights_host (n,).  n <= K. */
int bn_mppi_get_top_samples(bn_mppi_t *h, int32_t instance, int32_t n, float *states_host,
                            float *weights_host);

/* Rows of the LATEST solve's _state_seq_batch regenerated on the device (works in every mode but sampled-slip; it is how
 * lean mode serves get_top_samples, mppi.py:232-238): rollouts idx_device[0..n) (NULL: rollouts 0..n-1) -> out_device
 * (n,T+1,3), bit-identical to the rows a full-API solve stores (same noise, mean, start state, device functions).
 * Writes the pending tail first; with injected noise the caller's eps block of that solve must still be alive.
 * BN_ERR_STATE after a bn_mppi_set_map that followed the latest solve: the rows would be rolled out on the NEW map, not be that
 * solve's rollouts (in lean mode this also applies to bn_mppi_get_states / bn_mppi_get_top_samples). */
int bn_mppi_reroll_async(bn_mppi_t *h, int32_t instance, const int32_t *idx_device, int32_t n, float *out_device);

/* Zero-copy access to a library-owned device buffer (layouts in bn_buffer_id). */
int bn_mppi_device_buffer(bn_mppi_t *h, bn_buffer_id id, void **device_ptr, size_t *bytes);

/* Row pitch Kp (in floats) of BN_BUF_STATES / BN_BUF_CONTROLS. */
int32_t bn_mppi_row_pitch(const bn_mppi_t *h);

/* Number of solves enqueued so far (the Philox stream position). */
uint64_t bn_mppi_solve_count(const bn_mppi_t *h);

/* 0: the default arithmetic (carried heading vector, fused transit); 1: the reference's operation order
 * (BN_FLAG_REFERENCE_ORDER, or selected because dt * max|omega| > 0.5).  Kernel launches one solve costs: 1 or 2. */
int32_t bn_mppi_arithmetic(const bn_mppi_t *h);
int32_t bn_mppi_launches_per_solve(const bn_mppi_t *h);
/* Kernel launches of one bn_mppi_forward_async / bn_mppi_forward_state_async / bn_mppi_solve call (solve AND its tail): 1 on the
 * latency kernel (see bn_mppi_forward_state_async), else 2. */
int32_t bn_mppi_launches_per_forward(const bn_mppi_t *h);
/* How the cell index divides by the resolution (grid_map.py:203): 2 = exact multiplication (power-of-two resolution), 1 = the
 * three-instruction correctly rounded quotient, validated exhaustively on the device at create for this resolution and these limits
 * (bn_mppi_create refuses a resolution that fails the check; none is known). */
int32_t bn_mppi_fast_quotient(const bn_mppi_t *h);

/* With BN_FLAG_PROFILE: mean duration in milliseconds of the rollout kernel and of
 * the finish kernel over the solves since the last call (HIP events on the
 * handle's stream), and how many solves that covers.  Synchronises.
 * Two-launch mode: an event pair around every kernel.  Pipelined mode (one ~15 us
 * launch per solve, back to back): one event per group of 10 launches, mean = group
 * time / 10 (an event pair per launch would add its own dispatch latency);
 * finish_ms is 0 there -- the tail rides inside the next launch. */
int bn_mppi_kernel_ms(bn_mppi_t *h, float *rollout_ms, float *finish_ms, int32_t *n_solves);

/* Algorithmic HBM bytes of one solve in the current mode (DESIGN.md "Roofline"). */
int64_t bn_mppi_algorithmic_bytes(const bn_mppi_t *h, bn_noise_kind noise);
/* ... with the map term (4 G^2) replaced by the reachable window the kernels stage (4 WN^2): the bytes a solve has to move when
 * only a corner of the map can be reached within the horizon.  The roofline fraction of a lean launch against THIS figure is the
 * honest one (SURVEY 8d's lean formula counts the whole map). */
int64_t bn_mppi_algorithmic_bytes_window(const bn_mppi_t *h, bn_noise_kind noise);

/*
 * TraversabilityModel._infer_risk_map, traversability_model.py:28-51 (runs once per dynamics object, in
 * UnicycleModel.__init__): the (G,G) risk map from the predicted slip distribution Normal(mean, std).
 *   BN_RISK_EXPECTED  mean                                             (:30-31)
 *   BN_RISK_VAR       torch.quantile(samples, confidence, dim=0)       (:33-36), linear interpolation
 *   BN_RISK_CVAR      nanmean of the samples strictly above that       (:37-42)
 * samples_i = mean + std * z_i; z = (num_samples, G, G) standard normals supplied by the caller (the
 * reference's Normal.sample stream, for parity) or NULL: generated in the kernel (Philox, keyed by seed).
 * Stateless; host-resident arguments are copied and the call synchronises, device-resident ones are only
 * enqueued on `stream`.  Errors: bn_risk_last_error().
 */
typedef enum bn_risk_metric { BN_RISK_EXPECTED = 0, BN_RISK_VAR = 1, BN_RISK_CVAR = 2 } bn_risk_metric;
int bn_risk_map_infer(int32_t device_id, void *stream, const float *mean, const float *std, bn_mem_kind where_in,
                      int32_t grid_size, bn_risk_metric metric, float confidence, int32_t num_samples,
                      const float *z, bn_mem_kind where_z, uint64_t seed, float *out, bn_mem_kind where_out);
const char *bn_risk_last_error(void);

/* Test hook: the library's device arithmetic (DESIGN.md "Arithmetic spec") applied elementwise to n device floats:
 * fn 0 = correctly rounded sqrt, 1 / 2 = sin / cos of the spec, 3 = heading wrap (theta + pi) % 2pi - pi with
 * torch.remainder semantics (robot_model.py:90), 4 = its in-loop form, 5 = the sqrt for zero / normal finite arguments.  Lets the tests compare the kernels' building
 * blocks with the oracle's one value at a time. */
int bn_device_math_eval(int32_t fn, const float *in_device, float *out_device, int64_t n, void *stream);

const char *bn_last_error(void);
int bn_mppi_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BENCHNAV_MPPI_H */
