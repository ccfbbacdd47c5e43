/*
 * irbpp.h -- C ABI of the MI355X-native batched packing environment (libirbpp_hip.so).
 *
 * The reference (alexfrom0815/IR-BPP) has no FFI: its environment is Python behind the
 * VecEnv protocol (wrapper/vec_env.py:29-138, wrapper/shmem_vec_env.py:20-157).  This
 * header is the boundary a maintainer binds instead of spawning one process per bin:
 * each entry point names the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes stub.
 *
 * Conventions
 *   - every function returns 0 (IRBPP_OK) or a negative irbpp_status; no exceptions cross
 *     the ABI; irbpp_status_string() explains a code.
 *   - "_dev" pointers are caller-owned device (HBM) memory; everything else is host memory.
 *   - launches are asynchronous on the given hipStream_t (passed as void*; NULL = the null
 *     stream); a handle is not thread-safe; one outstanding step per handle (the
 *     reference's `waiting_step` rule, shmem_vec_env.py:58-74).
 *   - observation buffers are written in full on every call, so the caller may hand a
 *     fresh buffer each step (trainer.py:184-186 keeps the previous `state` alive); a buffer
 *     handed over with irbpp_register_obs_buffer is kept complete by the library instead
 *     (same contents, fewer stores).
 *
 * Limits (irbpp_create returns IRBPP_ERR_ARG beyond them): action grid <= 32 x 32 cells (up to 16 x 16: the tuned pipeline;
 * 17 .. 32 a side, e.g. resolutionA = 0.01: the capacity path of csrc/irbpp_wide.hip, one kernel per observation, heightmap
 * <= 64 x 64 cells, no stability proxy, no stage-level tooling entry points), heightmap <= 128 x 128 cells with resolutionA an
 * integer multiple of resolutionH and the bin an integer number of action cells; n_rot <= 8; selected <= 1024; buffer_size <= 16;
 * bin[2] / resolution_z <= 31 height levels (cvTools.py:78 codes a level in 6 bits: level + 32);
 * num_bins <= 1048576 per device.  Item ids are < 65536 in the placement log.
 */
#ifndef IRBPP_H
#define IRBPP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct irbpp_env irbpp_env;

typedef enum {
    IRBPP_OK = 0,
    IRBPP_ERR_ARG = -1,        /* bad argument / unsupported configuration          */
    IRBPP_ERR_HIP = -2,        /* a HIP runtime call failed                          */
    IRBPP_ERR_STATE = -3,      /* call order (e.g. step before load_shapes/reset)    */
    IRBPP_ERR_DEVICE = -4,     /* a kernel raised its error word (see irbpp_device_error) */
    IRBPP_ERR_NOMEM = -5
} irbpp_status;

/* Geometry and episode parameters: PackingGame.__init__ (binPhy.py:22-116),
 * Space.__init__ (space.py:15-47), Interface.__init__ (Interface.py:33-40). */
typedef struct {
    int32_t num_bins;        /* bins simulated on this device                                   */
    int32_t n_rot;           /* ZRotNum (arguments.py:117), 1..8                                 */
    int32_t selected;        /* selectedAction S (arguments.py:79), 1..1024                      */
    int32_t buffer_size;     /* bufferSize k (arguments.py:75); >1 = hierarchical, 1..16         */
    double  resolution_a;    /* resolutionA                                                      */
    double  resolution_h;    /* resolutionH                                                      */
    double  resolution_z;    /* resolutionZ = heightResolution (arguments.py:84,125)             */
    double  bin[3];          /* np.round([0.32,0.32,0.30], 6) (arguments.py:115)                 */
    double  scale_z;         /* Interface scale[2] = 100 (arguments.py:123)                      */
    int32_t traj_start;      /* trajectory of global bin 0's first episode; 1 = LoadItemCreator
                                (IRcreator.py:86-92 increments before first use)                */
    int32_t global_offset;   /* global index of this device's bin 0 (multi-GPU sharding)         */
    int32_t global_bins;     /* bins over all ranks = trajectory stride between episodes         */
    int32_t device;          /* HIP device ordinal                                               */
    int32_t stability;       /* stand-in for the rigid-body settling the path leaves out (Interface.py:271-310):
                                0 off; 1 every accepted placement is also rated by a static support test
                                (irbpp_step_out::stable_dev), results otherwise unchanged; 2 a placement that
                                fails the test is refused like one that does not fit (episode ends)          */
    int32_t tuning;          /* bit flags, none of which changes a result (A/B measurements and the parity test that
                                plays lattice data through both overlap paths): IRBPP_TUNE_*; 0 = the library decides */
    int32_t item_stream;     /* 0: trajectories (LoadItemCreator, IRcreator.py:74-103): episode e of global bin g reads row
                                (traj_start + g + e*global_bins) % n_traj from its start.  1: one endless stream per bin
                                (RandomItemCreator / RandomInstanceCreator / RandomCateCreator, IRcreator.py:26-72): bin b
                                of this device reads row b % n_traj and a new episode goes on where the last one stopped
                                (ItemCreator.reset only clears the queue); the row is a ring the host refills
                                (irbpp_stream_cursors)                                                          */
} irbpp_config;

#define IRBPP_TUNE_NO_BLOCK_PATH  1   /* lattice data (BlockOut) through the generic overlap test too               */
#define IRBPP_TUNE_WIDE_KERNEL    2   /* transition kernel without the 64-VGPR cap                                  */
#define IRBPP_TUNE_NARROW_KERNEL  4   /* transition kernel under the 64-VGPR cap even on the generic path           */
#define IRBPP_TUNE_NO_BOX_PATH    8   /* box data (Cube) through the generic overlap test too                       */
#define IRBPP_TUNE_NO_ITEM_ORDER 16   /* launch the bins in index order instead of grouped by observed item per XCD */
#define IRBPP_TUNE_TRACE_CPW64   32   /* border following with 64 / 32 / 16 candidate starts per wave whatever the number of */
#define IRBPP_TUNE_TRACE_CPW32   64   /* bins (default: by the number of bins, see launch_group in irbpp_capi.hip)          */
#define IRBPP_TUNE_TRACE_CPW16  128
#define IRBPP_TUNE_INLINE_POLYGON 256 /* every trace wave approximates the borders it followed itself; no polygon kernel (the
                                         path a full record list takes, forced for the parity tests; measured slower at every size) */
#define IRBPP_TUNE_NO_HEAVY_FIRST 512 /* emit kernel: the bins in launch order, speckled ones not first                        */
#define IRBPP_TUNE_FUSED_APPLY 2048 /* step(): the actions applied inside the transition kernel (one 256-thread workgroup per bin, as until
                                      round 4) whatever the size of the launch ...                                              */
#define IRBPP_TUNE_SPLIT_APPLY 4096 /* ... or by irbpp_apply_kernel (one wave per bin) in front of it whatever the size (default: split
                                      from the size on at which it pays, see split_apply in irbpp_capi.hip); identical results, for
                                      A/B runs and the parity tests                                                              */
#define IRBPP_TUNE_BLOCK_EMIT 8192 /* emit kernel: one 256-thread workgroup per bin also for lattice / box data (default there, from 2048
                                     bins per launch on: one wave per bin, four bins per workgroup); identical results          */
#define IRBPP_TUNE_WAVE_EMIT 16384 /* ... or the wave-per-bin form for lattice / box data whatever the size of the launch      */
#define IRBPP_TUNE_GRAPH 32768 /* step() / get_action_candidates() replayed as HIP graphs owned by the library (one per distinct argument
                                  set, captured at its second use) instead of launched kernel by kernel.  Off by default: slower on
                                  ROCm 7.2 at every size measured (launch_env in irbpp_capi.hip); identical results                 */
#define IRBPP_TUNE_WG512 65536 /* generic overlap path: the transition kernel with 512-thread workgroups (eight waves share a bin's tile)
                                  whatever the LDS per bin; default: where at most four 256-thread workgroups fit a CU's LDS      */
#define IRBPP_TUNE_NO_WG512 131072 /* ... never                                                                                 */
#define IRBPP_TUNE_TRACE_REFILL 262144 /* border following in batches of 128 candidate starts per wave whose lanes take the next candidate as
                                         they close their borders.  Off by default: measured slower at every size (profiles/r06/LOG.md,
                                         session 2); identical results, parity-tested                                                       */
#define IRBPP_TUNE_NO_MIXED_PATH 524288 /* a data set of which only SOME rotations are lattice footprints (BlockOut at eight rotations) through
                                          the cell lists entirely, as until round 5 (default: those rotations on the block path, the others on
                                          their lists, in one kernel); identical results                                                    */
#define IRBPP_TUNE_CHAIN 1048576     /* every observation as ONE kernel: the bin's workgroup also follows its borders, approximates them and
                                        writes the candidate rows (no trace / polygon / emit launches).  Off by default: measured slower at
                                        the launch sizes it was built for (irbpp_capi.hip: chain_launch); identical results, parity-tested  */
#define IRBPP_TUNE_WG128 2097152     /* BlockOut at R = 4: the transition kernel with 128-thread workgroups (two waves per bin, two action cells
                                        per thread); identical results, for A/B runs and the parity tests                                     */
#define IRBPP_TUNE_RECT 4194304      /* the transition kernel marks the vertices of level-image components that are isolated solid rectangles
                                      * itself, as it does isolated pixels (default: the trace kernel follows them like every other border --
                                      * measured: trace -5 us, polygon -2 us, transition +6 us at 8192 BlockOut bins, profiles/r06/LOG.md s21) */
#define IRBPP_TUNE_NO_SPECIALISED 1024 /* the run-time builds of the transition / emit kernels even where a build with the
                                         geometry as compile-time constants exists (16 x 16 action cells, step 2 or 4, R = 2 / 4 / 8,
                                         S = 500: BASELINE.json's configs); identical results, for A/B runs and the parity tests  */

/* Per-step outputs beyond the observation: what PackingGame.step returns and what Monitor
 * adds on `done` (binPhy.py:299-311,327; monitor.py:58-75).  All device pointers, one entry
 * per bin; any pointer may be NULL to skip that output. */
typedef struct {
    double*  reward_dev;      /* reward of this step (item_ratio*10 or 0.0)                      */
    uint8_t* done_dev;        /* 1 iff the episode ended with this step                          */
    int32_t* counter_dev;     /* info['counter'] = items packed, valid where done                */
    double*  ratio_dev;       /* info['ratio']   = get_ratio(),  valid where done                */
    double*  ep_reward_dev;   /* sum of the episode's rewards (Monitor 'r' before round(.,6))    */
    int32_t* ep_len_dev;      /* Monitor 'l'                                                     */
    uint8_t* stable_dev;      /* irbpp_config::stability >= 1: 1 iff the placement of this step rests stably
                                 (0 also where the step placed nothing); NULL to skip                    */
    int32_t* err_dev;         /* copy of the device error word after this step (one int32, not per bin):
                                 callers that fetch the outputs with one D2H copy get it for free.
                                 LIFETIME: the library keeps this pointer after irbpp_step returns and ORs
                                 error bits into the word as later launches of this environment raise them
                                 (irbpp_get_action_candidates writes through it): it must stay allocated and
                                 must not be written by the caller until the next irbpp_step with another
                                 err_dev (or NULL), irbpp_reset / irbpp_reset_bins, or irbpp_destroy.  The word
                                 always holds every sticky bit of the device error word when a step's
                                 kernels have run (it is re-seeded whenever the pointer changes and after
                                 every reset / stream write)                                          */
} irbpp_step_out;

const char* irbpp_status_string(int status);
int irbpp_version(void);
/* Hash of the sources and compiler flags this binary was built from (irbpp_amd/build.py: source_hash(), passed in as
 * -DIRBPP_SOURCE_HASH); "unstamped" for a build made some other way.  The Python loader refuses a binary whose stamp is
 * not the hash of the sources lying next to it: a stale .so can then never be measured or tested by mistake. */
const char* irbpp_source_hash(void);

/* replaces: gym.make('Physics-v0', args=args) x num_processes + ShmemVecEnv.__init__
 * (envs.py:67-99, shmem_vec_env.py:25-59) */
int irbpp_create(const irbpp_config* cfg, irbpp_env** out);
/* replaces: ShmemVecEnv.close_extras (shmem_vec_env.py:83-92) */
int irbpp_destroy(irbpp_env* env);

/* replaces: args.shotInfo / args.infoDict built by shotInfoPre + load_shape_dict
 * (tools.py:227-279).  Shape id k, rotation r owns four [fx,fy] row-major float64 tables
 * starting at pool offset offsets[k*n_rot+r]; dims[(k*n_rot+r)*2 + {0,1}] = fx, fy;
 * extents[(k*n_rot+r)*3 + {0,1,2}] = mesh.extents; volumes[k] = infoDict[k][0]['volume'].
 * Masks must be exactly 0.0 or 1.0.  All host pointers; copied. */
int irbpp_load_shapes(irbpp_env* env, int32_t n_shapes,
                      const double* extents, const double* volumes,
                      const int32_t* dims, const int64_t* offsets, int64_t pool_len,
                      const double* height_top, const double* height_bottom,
                      const double* mask_top, const double* mask_bottom);

/* replaces: LoadItemCreator.item_trajs = torch.load(test_sequence.pt) (IRcreator.py:81) and
 * the pre-drawn np.random.choice stream of the Random*Creator classes (IRcreator.py:26-72).
 * ids: int32[n_traj][length] host memory, copied. */
int irbpp_load_sequences(irbpp_env* env, const int32_t* ids, int32_t n_traj, int32_t length);

/* obs_len of PackingGame (binPhy.py:87-98): which=0 the observation reset()/step() return
 * (5S+9+Hx*Hy when k==1, k+Hx*Hy when k>1); which=1 the location observation of
 * get_action_candidates (5S+9+Hx*Hy). */
int irbpp_obs_len(const irbpp_env* env, int32_t which);

/* replaces: ShmemVecEnv.reset (shmem_vec_env.py:61-68) -> PackingGame.reset (binPhy.py:128-147).
 * obs_dev: float32[num_bins][obs_len(0)].  Restarts the trajectory counters. */
int irbpp_reset(irbpp_env* env, float* obs_dev, void* stream);

/* replaces: ShmemVecEnv.reset_specific (shmem_vec_env.py:113-117): PackingGame.reset of the
 * listed bins only.  bins_dev: int32[count] distinct local bin indices (device memory);
 * obs_dev: float32[count][obs_len(0)], row i = the reset observation of bin bins_dev[i].
 * Like the reference's per-env reset the bin moves on to its next trajectory and the episode it
 * abandons enters no statistics.  An index outside [0, num_bins) is skipped and raises
 * IRBPP_DEVERR_BAD_BIN. */
int irbpp_reset_bins(irbpp_env* env, const int32_t* bins_dev, int32_t count, float* obs_dev, void* stream);

/* replaces: ShmemVecEnv.step_async+step_wait (shmem_vec_env.py:70-81) -> PackingGame.step
 * (binPhy.py:248-337, no-physics branch) + the worker's auto-reset (shmem_vec_env.py:141-144).
 * actions_dev: int32[num_bins] indices into the candidate rows of the last location
 * observation.  obs_dev: float32[num_bins][obs_len(0)]; for a finished episode it already
 * holds the next episode's first observation. */
int irbpp_step(irbpp_env* env, const int32_t* actions_dev, float* obs_dev,
               const irbpp_step_out* out, void* stream);

/* replaces: ShmemVecEnv.get_action_candidates (shmem_vec_env.py:99-102) ->
 * PackingGame.get_action_candidates (binPhy.py:161-169).  order_actions_dev: int32[num_bins]
 * buffer slots; loc_obs_dev: float32[num_bins][obs_len(1)].  Hierarchical mode only. */
int irbpp_get_action_candidates(irbpp_env* env, const int32_t* order_actions_dev,
                                float* loc_obs_dev, void* stream);

/* replaces: PackingGame.get_all_possible_observation (binPhy.py:171-180; no caller in the reference): the location
 * observation of EVERY buffer slot of every bin on the current heightmaps, loc_obs_dev: float32[num_bins][k][obs_len(1)]
 * -- per bin the concatenation the reference returns.  Hierarchical mode only.  As there, the candidate rows a following
 * irbpp_step would index are those of the LAST slot, and the slot chosen by the last irbpp_get_action_candidates
 * (self.orderAction) is left as it is.  Ignores irbpp_set_auto_policy's buffer semantics only in that the action written is
 * the last slot's. */
int irbpp_get_all_possible_observation(irbpp_env* env, float* loc_obs_dev, void* stream);

/* The scripted policy used by the benchmark and the parity tests: per bin the candidate
 * row with V==1 and the lowest H (first on ties), 0 if none.  loc_obs_dev has row stride
 * obs_stride floats.  (Stands in for agent.py:51-58 so no host round trip is timed.) */
int irbpp_policy_minz(irbpp_env* env, const float* loc_obs_dev, int32_t obs_stride,
                      int32_t* actions_dev, void* stream);

/* The same policy fused into the observation: while actions_dev is set (int32[num_bins] on the device; NULL
 * switches it off), every call that emits a location observation -- irbpp_reset, irbpp_reset_bins (listed bins
 * only), irbpp_step of an online environment, irbpp_get_action_candidates -- also writes the action
 * irbpp_policy_minz would pick on it, so that a scripted roll-out (tools.test with a heuristic, the benchmark)
 * needs no policy kernel between two steps.  The buffer may be the one the next irbpp_step reads its actions from. */
int irbpp_set_auto_policy(irbpp_env* env, int32_t* actions_dev);

/* Registers a location-observation buffer (float32[num_bins][obs_len(1)], i.e. what irbpp_reset / irbpp_step of an
 * online environment / irbpp_get_action_candidates write) that from now on ONLY this library writes: it remembers
 * how many candidate rows each bin's block holds, so a later call that is handed the same pointer stores the rows
 * that exist and clears the ones that existed before, instead of rewriting the zero tail of all `selected` rows
 * (typically 80 % of the block).  The contents delivered are the same as for an unregistered buffer.  Up to 8
 * buffers (a ring of 2-3 is what an actor loop needs); not for irbpp_reset_bins.
 * Lifetime: the registration is keyed by the pointer value, so unregister a buffer BEFORE freeing it (a later
 * allocation may reuse the address); registering a pointer again, irbpp_invalidate_obs_buffer, and any library call
 * that writes another layout through it (irbpp_step of a buffered environment) make the next emit rewrite all
 * `selected` rows.  If the caller itself writes into a registered buffer it must invalidate it afterwards. */
int irbpp_register_obs_buffer(irbpp_env* env, float* obs_dev);
int irbpp_unregister_obs_buffer(irbpp_env* env, float* obs_dev);
/* obs_dev == NULL: every registered buffer.  Asynchronous on `stream`. */
int irbpp_invalidate_obs_buffer(irbpp_env* env, float* obs_dev, void* stream);

/* item_stream = 1 only.  The position of every bin in its stream row, counted in items consumed since the row was
 * loaded (int32[num_bins], device memory): set == 0 reads them, set != 0 writes them.  The host side of the ring
 * (irbpp_amd/itemgen.py) reads the cursors, rewrites the consumed part of each row with irbpp_stream_write and never
 * lets a bin run more than `length` items ahead of what it has written. */
int irbpp_stream_cursors(irbpp_env* env, int32_t* cursors_dev, int32_t set, void* stream);
/* Row r of the stream table gets ids_dev[r][0 .. count_dev[r]) (int32[n_traj][width], device memory) at ring positions
 * (first_dev[r] + c) mod length.  Ids below -1 are stored as -1 (no item): the value -3 is the bins' own mark of a slot
 * they have consumed. */
int irbpp_stream_write(irbpp_env* env, const int32_t* ids_dev, const int32_t* first_dev, const int32_t* count_dev,
                       int32_t width, void* stream);

/* replaces: RandomItemCreator / RandomInstanceCreator / RandomCateCreator.generate_item (IRcreator.py:26-72) of ONE
 * environment whose process was seeded with `seed` (= args.seed + rank, envs.py:41 -> binPhy.py:118-123): host-side,
 * the same MT19937 words, masks and rejections as np.random.choice on the legacy global generator, so the stream of
 * item ids equals the reference's.  n_groups > 0: two stages, name = choice(n_groups), item = choice(members of that
 * group) with group g = members[group_offsets[g] .. group_offsets[g+1]) in the order the reference's dict holds them;
 * n_groups == 0: one stage over `members` (RandomItemCreator).  irbpp_itemgen_draw appends `count` items. */
typedef struct irbpp_itemgen irbpp_itemgen;
int irbpp_itemgen_create(uint32_t seed, int32_t n_groups, const int32_t* group_offsets, const int32_t* members,
                         int32_t n_members, irbpp_itemgen** out);
int irbpp_itemgen_draw(irbpp_itemgen* gen, int32_t count, int32_t* out_host);
int irbpp_itemgen_destroy(irbpp_itemgen* gen);

/* -- stage-level entry points (parity tests and tooling) ------------------------------- */

/* Space.get_possible_position (space.py:98-129) for item_ids_dev[b] on bin b's current
 * heightmap.  posz_dev: float64[num_bins][n_rot][Ax][Ay] = posZmap; mask_dev:
 * uint8[num_bins][n_rot][Ax][Ay] = naiveMask.  Does not modify the environment. */
int irbpp_possible_position(irbpp_env* env, const int32_t* item_ids_dev,
                            double* posz_dev, uint8_t* mask_dev, void* stream);

/* Space.get_heuristic_action (space.py:162-218) for the item of the last location observation on
 * the current heightmaps: method 1 MINZ, 2 DBLF, 3 FIRSTFIT, 4 HM; dir_idx 0..3 = (Xflip, Yflip)
 * as at space.py:163-166.  out_dev: int32[num_bins][3] = (rotIdx, lx, ly), the first minimum in C
 * order of np.round(score, 6) with invalid cells at 1e6.  RANDOM is not provided. */
int irbpp_heuristic_action(irbpp_env* env, int32_t method, int32_t dir_idx, int32_t* out_dev, void* stream);

/* replaces: tools.shot_item (tools.py:98-135), the per-(shape, rotation) footprint precompute that
 * shotInfoPre caches on disk (tools.py:248-279) -- trimesh ray casting in the reference, a
 * z-ray/triangle rasteriser here.  Stateless.  verts_dev: float64[n_verts][3] of the mesh already
 * rotated and translated to its bounding-box minimum; faces_dev: int32[n_faces][3]; ray (i, j)
 * passes through (i*resolution_h + shift, j*resolution_h + shift) (shift = 0.001, tools.py:81-95).
 * Outputs float64[fx][fy] each: heightMapT (highest hit), heightMapB (lowest hit), maskH, maskB;
 * if no ray hits at all: T = extent_z, B = 0, masks 1 (tools.py:112-117,126-131).
 * scratch_dev: one int32 of device memory. */
int irbpp_shot_item(const double* verts_dev, const int32_t* faces_dev, int32_t n_faces, int32_t fx, int32_t fy,
                    double resolution_h, double shift, double extent_z, double* top_dev, double* bottom_dev,
                    double* mask_top_dev, double* mask_bottom_dev, int32_t* scratch_dev, void* stream);

/* getConvexHullActions (cvTools.py:61-102) on caller-supplied grids, independent of the
 * environment state: posz_valid_dev float64[n_grids][n_rot][Ax][Ay], mask_dev
 * uint8[n_grids][n_rot][Ax][Ay].  vertex_rows_dev: uint32[n_grids][n_rot][16]; word `row` has
 * bit `col` set iff (row, col) is a candidate of that rotation (np.unique makes the candidate
 * list a set, cvTools.py:101, so the bit grid is its exact representation). */
int irbpp_convex_hull_actions(irbpp_env* env, int32_t n_grids, const double* posz_valid_dev,
                              const uint8_t* mask_dev, uint32_t* vertex_rows_dev, void* stream);

/* Space.heightmapC of every bin (space.py:26): float64[num_bins][Hx][Hy]. */
int irbpp_get_heightmaps(irbpp_env* env, double* hm_dev, void* stream);
/* (the drop heights a step takes from the last observation are forgotten with the old maps: a step that follows
 * irbpp_set_heightmaps without a new observation recomputes its drop height on the new map) */
int irbpp_set_heightmaps(irbpp_env* env, const double* hm_dev, void* stream);

/* Running totals over finished episodes since create/reset, for logging
 * (trainer.py:215-222): out_dev float64[4] = {episodes, sum ratio, sum counter, sum reward}.
 * The multi-GPU runner all-reduces these four numbers (RCCL). */
int irbpp_episode_totals(irbpp_env* env, double* out_dev, void* stream);

/* replaces: PackingGame.packed (binPhy.py:141,296), the per-episode placement record that
 * tools.test saves to trajs.npy (tools.py:339-340).  While set, every successful placement of
 * bin b, the i-th of its episode (i < capacity), stores meta_dev[b*capacity+i] =
 * item | rot<<16 | lx<<20 | ly<<24 (item ids must be < 65536) and z_dev[b*capacity+i] = its drop
 * height posZmap[rot,lx,ly].  A finished episode's entries stay readable until that bin's next
 * placement overwrites them, so read them right after the step that reported done.  NULL, NULL
 * switches the log off. */
int irbpp_set_placement_log(irbpp_env* env, uint32_t* meta_dev, double* z_dev, int32_t capacity);

/* Caller side: the N per-env prioritised replay memories of main.py:61-63 as one tensor set.
 * replaces: SegmentTree.find/_retrieve (memory.py:72-86) for `draws` values per env.
 * tree_dev float32[n_env][2*capacity-1] (implicit heap, leaves at capacity-1..), values_dev float32[n_env][draws];
 * outputs [n_env][draws]: leaf value, data index (tree index - capacity + 1), tree index. */
int irbpp_sumtree_find(const float* tree_dev, int32_t n_env, int32_t capacity, const float* values_dev, int32_t draws,
                       float* prob_dev, int64_t* data_idx_dev, int64_t* tree_idx_dev, void* stream);
/* ReplayMemory._get_samples_from_segments (memory.py:161-176) for all envs: draws one position per (env, segment)
 * -- `draws` segments of p_total/draws each -- with the rejection loop of memory.py:170-176 run on the device
 * (redraw while the position straddles the write index index_dev[env] +- n_step / history or has probability 0).
 * Counter-based uniform numbers from `seed`; failed_dev[0] |= 1 if a draw found nothing valid in max_tries tries. */
int irbpp_sumtree_sample(const float* tree_dev, const int64_t* index_dev, int32_t n_env, int32_t capacity, int32_t draws,
                         int32_t n_step, uint64_t seed, int32_t max_tries, float* prob_dev, int64_t* data_idx_dev,
                         int64_t* tree_idx_dev, int32_t* failed_dev, void* stream);

/* replaces: SegmentTree.update/_propagate (memory.py:47-58) for `leaves` (tree index, value) pairs per env, applied
 * in list order, every ancestor recomputed as left + right in float32; max_dev float32[n_env] is SegmentTree.max.
 * env_mask_dev (may be NULL) uint8[n_env]: envs with 0 are skipped.  IRBPP_ERR_ARG if 2*capacity-1 > 16384 (the
 * caller then keeps its own path). */
int irbpp_sumtree_update(float* tree_dev, float* max_dev, int32_t n_env, int32_t capacity, const int64_t* tree_idx_dev,
                         const float* priority_dev, int32_t leaves, const uint8_t* env_mask_dev, void* stream);
/* The replay tensors of all envs (memory.py:28-36 per env): float32 states [n_env][capacity][obs_len], int64 actions,
 * float32 rewards, uint8 nonterminals [n_env][capacity], float32 tree [n_env][2*capacity-1], int64 index and uint8 full
 * [n_env], float32 scaling [n_step] = discount^k. */
typedef struct {
    const float* states_dev; const int64_t* actions_dev; const float* rewards_dev; const uint8_t* nonterminals_dev;
    const float* tree_dev; const int64_t* index_dev; const uint8_t* full_dev; const float* scaling_dev;
    int32_t n_env, capacity, obs_len, n_step;
} irbpp_replay_view;
/* replaces: ReplayMemory._get_transition_new and the batch assembly of ReplayMemory.sample (memory.py:123-139,178-204)
 * for `draws` (<= 256) positions per env found by irbpp_sumtree_find: outputs env-major rows [n_env*draws]:
 * state and next state float32 [..][obs_len], action int64, n-step return, non-terminal flag, importance weight
 * (priority_weight = beta) float32. */
int irbpp_replay_gather(const irbpp_replay_view* view, int32_t draws, float beta, const int64_t* data_idx_dev,
                        const float* prob_dev, float* state_dev, int64_t* action_dev, float* return_dev,
                        float* next_state_dev, float* nonterminal_dev, float* weight_dev, void* stream);
/* The writable side of the same tensor set (memory.py:28-36, 111): int32 timesteps [n_env][capacity], float32 max priority,
 * int32 episode timestep counter [n_env]; the others as in irbpp_replay_view. */
typedef struct {
    float* states_dev; int64_t* actions_dev; float* rewards_dev; uint8_t* nonterminals_dev; int32_t* timesteps_dev;
    float* tree_dev; float* max_dev; int64_t* index_dev; uint8_t* full_dev; int32_t* t_dev;
    int32_t n_env, capacity, obs_len;
} irbpp_replay_store;
/* replaces: ReplayMemory.append (memory.py:117-121) with SegmentTree.append (:60-70) and its _propagate (:47-52), called
 * per env at trainer.py:184-186 -- for all envs in one launch: state_dev float32 rows `state_stride` floats apart, action
 * int32 or int64 (action_bytes 4 | 8), reward float32 or float64 (reward_bytes 4 | 8; stored as float32), terminal_dev
 * uint8[n_env]; valid_dev (may be NULL = every env) uint8[n_env]: envs with 0 are skipped (a sample that is not Valid). */
int irbpp_replay_append(const irbpp_replay_store* store, const float* state_dev, int64_t state_stride, const void* action_dev,
                        int32_t action_bytes, const void* reward_dev, int32_t reward_bytes, const uint8_t* terminal_dev,
                        const uint8_t* valid_dev, void* stream);
/* replaces: the tail of Agent.act (agent.py:55-58) with get_mask_from_state (tools.py:298-299) fused in:
 * action[e] = argmax_i q[e][i] over the candidates i whose validity flag obs[e][5*i+4] is non-zero (first maximum;
 * 0x7fffffff never occurs: with no valid candidate every q is -inf and index 0 wins, like torch.argmax). */
int irbpp_masked_argmax(const float* q_dev, int32_t q_stride, const float* obs_dev, int32_t obs_stride, int32_t selected,
                        int32_t n_env, int64_t* action_dev, void* stream);

/* Tooling: when cycles_dev != NULL every later transition launch stores, per bin, one row
 * int64[num_bins][16]: shader-clock stamps 0 start, 1 action applied, 2 overlap test done,
 * 3 contour stage done, 4 observation written; 5..7 contour-stage detail; 8/9 the 100 MHz wall
 * clock at entry/exit; 10 = HW_ID | XCC_ID<<32 of the CU that ran the bin; 11..15 unused.
 * NULL switches it off. */
int irbpp_debug_phase_cycles(irbpp_env* env, int64_t* cycles_dev);

/* Tooling: time the transition kernel alone.  capacity > 0 creates a ring of that many HIP event
 * pairs; every later transition launch (reset / step / get_action_candidates) records the next
 * pair on its stream right around irbpp_env_kernel (the few-microsecond ordering kernel in front
 * of it stays outside the bracket).  capacity 0 frees the ring.
 * irbpp_debug_kernel_times waits for the recorded launches and writes the durations of the latest
 * min(max_count, recorded) of them in ms to ms_host, oldest first; *count says how many; the ring
 * is then empty again. */
/* Tooling: LDS bytes per workgroup of the transition kernel for this configuration and the names of the kernels a step over
 * all bins of this environment launches, " + "-separated, the transition kernel's build first (a string owned by the
 * environment, rewritten by the next call).  Valid after irbpp_load_shapes. */
int irbpp_debug_kernel_info(const irbpp_env* env, int32_t* lds_bytes, const char** kernel_name);
/* The overlap path irbpp_load_shapes chose for the data set: 1 block path (every footprint a union of uniform b x b
 * tiles: lattice data), 2 box path (every footprint a solid box), 3 generic cell lists, 4 mixed (the block path for the rotations
 * whose footprints are lattice footprints, cell lists for the others: BlockOut at eight rotations); IRBPP_ERR_STATE before the
 * shapes are loaded.  (vec_env.groups_for decides by it how many groups of bins to step a data set as.) */
int irbpp_overlap_path(const irbpp_env* env);
int irbpp_debug_kernel_timing(irbpp_env* env, int32_t capacity);
/* events around every `every`-th transition only (default 1): two event packets per step cost the stream ~5 % at
 * 0.15 ms per step, so bench.py samples every fourth step of its timed region */
int irbpp_debug_kernel_timing_every(irbpp_env* env, int32_t every);
int irbpp_debug_kernel_times(irbpp_env* env, float* ms_host, int32_t max_count, int32_t* count);

/* Device-side error word raised by kernels (0 = none).  Synchronises the stream. */
int irbpp_device_error(irbpp_env* env, void* stream, int32_t* flags_out);

#define IRBPP_DEVERR_LEVEL_RANGE   1   /* a height level fell outside the 64 supported bins  */
#define IRBPP_DEVERR_TRACE_GUARD   2   /* border following exceeded its iteration guard      */
#define IRBPP_DEVERR_BAD_ITEM      4   /* item id outside the loaded shape table             */
#define IRBPP_DEVERR_BAD_BIN       8   /* irbpp_reset_bins: bin index outside [0, num_bins)  */
#define IRBPP_DEVERR_CAPACITY     16   /* a die's candidate list overflowed (it holds twice the worst case of a fair
                                          share of the bins): results of that step are incomplete                */
#define IRBPP_DEVERR_BAD_ACTION   64   /* irbpp_step / irbpp_get_action_candidates: an action outside [-S, S) (order action: [-k, k)):
                                          the reference raises IndexError at binPhy.py:235 / :163; a negative index in
                                          range counts from the end, as there.  The step ran on a clamped index       */
#define IRBPP_DEVERR_STREAM_DRY   32   /* item_stream = 1: a bin fetched a ring slot it had consumed already and the host
                                          had not rewritten (irbpp_stream_write): its episode got no item there  */

#ifdef __cplusplus
}
#endif
#endif /* IRBPP_H */
