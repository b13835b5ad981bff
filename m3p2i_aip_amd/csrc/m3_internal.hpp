// m3_internal.hpp -- structures shared by the kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/m3p2i_hip.h"
#include "planar_dyn.hpp"
#include "point_cost.hpp"
#include "panda_dyn.hpp"

namespace m3 {

// ---- kernel argument blocks (passed by value; wave-uniform, live in SGPRs) -------------
struct RolloutArgs {
    int Kg, Kl, k0, T, nu;
    int multi_modal, mode_simple, sampling_random, sample_null_action;
    int gripper_cmd;
    float u_min[M3_MAX_NU], u_max[M3_MAX_NU], scale_tril[M3_MAX_NU], sigma_inv[M3_MAX_NU];
    float u_scale, gamma, lambda_;
    // the MPPIConfig switches no shipped config turns on (general kernel instance only)
    int noise_abs_cost, full_sigma;
    float noise_mu[M3_MAX_NU];
    const float* noise_mats;  // device [2][nu][nu]: Cholesky factor | inverse of noise_sigma (full_sigma)
    const float* scale_dev;   // device [nu] scale_tril when update_cov rewrites it every command, else null
    unsigned long long seed;
    unsigned call;
    CostParams cp;
    int lanes;                // active lanes (= samples) per 64-wide wavefront: 1..64
    float world0[18];         // rx ry rvx rvy | B: x y c s vx vy w | D: x y c s vx vy w
    // if sim_dof != null the initial world is read from env 0 of the wrapper's tensors
    // (dof_state [*,4], root_state [*,nA,13]) inside the kernel: no upload, no extra launch
    const float* sim_dof;
    const float* sim_root;
    int sim_box, sim_dyn;
    const float* delta;       // [T][Kl][nu], rows in lane-slot order when `order` is set
    const int* order;         // [Kl] lane -> local sample (null = identity)
    const float* mean;        // [T][nu] (U in simple mode)
    const float* mean1;
    const float* mean2;
    const float* best1;
    const float* best2;
    float* pend;              // [4][Kl]
    float* states;            // [T][Kl][4]
    float* actions;           // [T][Kl][nu]
    float* cost_h;            // [T][Kl]
    float* J;                 // [Kl]
    float* wave_min;          // [workgroups][3] minima of each workgroup's costs (wave_min.hpp), or null
};

struct SearchOut;
struct VI {  // (cost, global sample index) candidate of the top-k selection
    float v;
    int i;
};

struct UpdateArgs {
    VI* cand;         // [n_cand][M3_TOPK] per-workgroup top-k candidates (stage A -> stage B)
    float* part_min;  // [n_mins][3] per-workgroup minima (all, first half, second half)
    int n_mins;       // workgroups of k_mins
    int n_cand;       // top-k stage-A workgroups
    float* lad;       // [n_lad][96][3] per-workgroup eta sums on the beta ladders (k_ladder)
    int n_lad;        // workgroups of k_ladder
    struct SearchOut* srch;  // beta / eta / minima found by k_search (split path, K > 8192 multi-modal)
    float* apart;     // [apply_workgroups][8] partial half sums and argmax keys of k_apply_weights
    float* wpart;     // [n_chunk][3][T][nu] partial weighted sums of k_wsum (n_chunk > 1 only)
    int* wcount;      // [T] arrival counters of the k_wsum chunks + [T] the launch-wide one of the
                      // fused finalize (zero between launches)
    int fuse_finalize;  // k_wsum's last workgroup also does k_finalize's work (unsharded m3_command)
    int ladder_spins;   // k_update_small: bound of the wait for the other workgroups' ladder points
    int* lflag;         // [n_lad] k_ladder_search: epoch of the launch whose partial table the ladder workgroup has stored
    int epoch;
    int n_chunk;      // k_wsum workgroups per time step
    int lds_floats;   // costs staged in dynamic LDS by k_weights (set by launch_weights)
    int Kg, Kl, k0, T, nu;
    int multi_modal, mode_simple, env_type, filter_u, u_per_command;
    float lambda_, step_size_mean;
    const float* Jall;   // [Kg]
    float* w;            // [Kg]
    float* w1;           // [Kg/2]
    float* w2;           // [Kg - Kg/2]
    int* top_idx;        // [M3_TOPK]
    m3_info* info;       // device
    const float* actions;  // [T][Kl][nu]
    const float* states;   // [T][Kl][4]
    float* reduce;         // packed partial sums, see reduce_layout()
    float* mean;
    float* mean1;
    float* mean2;
    float* best;
    float* best1;
    float* best2;
    float* action_out;   // [T][nu]
    const int* p2p_err;  // device-side exchange attached: its sticky error word (a rank that never arrived: the records hold NaN)
    float* top_trajs;    // [M3_TOPK][T][2]
    // shard_mix (one-collective sharding): k_weights / top-k see the LOCAL shard (Kg = Kl,
    // Jall = local costs, w = weights + k0); kbase maps their sample numbers to global indices
    int kbase;           // global index of sample 0 of the costs k_weights sees (0 unless shard_mix)
    int half_g;          // global K/2 (mode split, pull preference)
    float* record;       // this rank's record (null unless shard_mix)
    const float* records_all;  // [n_ranks][record_len]
    int n_ranks, rank;
    float* top_dst;      // where top-k stage B writes the rows: top_trajs (unsharded) or the
                         // reduce buffer's top section (sharded: summed over ranks first)
    float* rec_topj;     // where topk_finish leaves the sorted top-k costs / indices for the other
    float* rec_topi;     //   ranks (inside this rank's record; null: nowhere)
    // ---- "regen" one-collective sharding (multi-modal): after ONE all-gather of records
    // {J of the shard | its top-k} every rank runs the whole update on all K samples; the actions of
    // the other ranks' samples are RE-GENERATED from the replicated noise table and plan instead of
    // being communicated (a_k[t] is a function of the global sample index: mppi.py:381-416)
    int regen;           // 1: the launch covers all K_global samples (Kl == Kg, k0 == 0 in this struct)
    int fast;            // shard_mix = 2: the records carry per-rank minima + ladder tables; the search mixes them
                         // instead of re-evaluating all K costs, and ONE kernel does weights + sums + finalize
    float* rec_mins;     // this rank's record: minima [4] and ladder table [LAD_N][3] (pre-gather launch)
    float* rec_table;
    int Kls;             // samples per shard (= per rank)
    int rec_len;         // floats per gathered record
    const float* noise_all;  // [n_ranks][T][Kls][nu]: every shard's noise block
    float* Jout;             // regen: k_mins (the first reader of the gathered records) leaves the costs
                             // here as one [K_global] array (== Jall of the later launches)
    float u_min[M3_MAX_NU], u_max[M3_MAX_NU], scale_tril[M3_MAX_NU];
    float u_scale;
    int sample_null_action, gripper_cmd;
    // ---- shard_mix = 3 (two small exchanges, O(K_local) work after the first): the second record of this rank,
    // {-w, index of its best sample per weight set (3 pairs) | its half sums (2)} + its weighted sums and best rows
    float* rec_b;              // k_apply_weights' combine leaves the 8-float header here (null: not this protocol)
    const float* recb_all;     // [n_ranks][recb_len] after the second exchange
    int recb_len;
};
// record B (shard_mix = 3): header | [3][T][nu] weighted sums of the LOCAL shard (globally normalised weights) |
// [3][T][nu] action rows of the local best samples
constexpr int RECB_HDR = 8;
__host__ __device__ inline int recb_length(int T, int nu) { return (RECB_HDR + 6 * T * nu + 3) / 4 * 4; }

// REDUCE buffer: [3][T][nu] weighted sums (all, mode 1, mode 2) | [3][T][nu] best rows
// (best, best_1, best_2; zero unless this rank owns the row) | [TOPK][T][2] top trajectories
__host__ __device__ inline int reduce_off_psum(int which, int T, int nu) { return which * T * nu; }
__host__ __device__ inline int reduce_off_best(int which, int T, int nu) { return (3 + which) * T * nu; }
__host__ __device__ inline int reduce_off_top(int T, int nu) { return 6 * T * nu; }
__host__ __device__ inline int reduce_length(int T, int nu) { return 6 * T * nu + M3_TOPK * T * 2; }
// shard_mix record: header | REDUCE-shaped body
//   [0] min cost m_r  [1] eta_r  [2],[3] half sums of the local weights  [4] best index (int bits)
//   [8..28) top-k costs  [28..48) top-k global indices (int bits)
constexpr int REC_HDR = 48, REC_TOPJ = 8, REC_TOPI = 28;
__host__ __device__ inline int record_length(int T, int nu) { return REC_HDR + reduce_length(T, nu); }
constexpr int MIX_MAX_RANKS = 32;
// beta ladders of the multi-modal search: 0.9^j (j = 0..63) and 1.2^j (j = 1..32), see update.hip
constexpr int LAD_S = 64, LAD_G = 32, LAD_N = LAD_S + LAD_G;
// regen record: [Kls] trajectory costs of the shard | [TOPK] its smallest costs | [TOPK] their global
// indices (int bits) | [TOPK][T][2] their (x, y) trajectories | [4] the shard's minima (all / mode 1 /
// mode 2 / pad) | [LAD_N][3] its eta(beta) sums on the ladders relative to those minima (the last two only
// with shard_mix = 2); length padded to a multiple of 4 floats
__host__ __device__ inline int regen_off_topj(int Kls) { return Kls; }
__host__ __device__ inline int regen_off_topi(int Kls) { return Kls + M3_TOPK; }
__host__ __device__ inline int regen_off_trajs(int Kls) { return Kls + 2 * M3_TOPK; }
__host__ __device__ inline int regen_off_mins(int Kls, int T) { return Kls + 2 * M3_TOPK + M3_TOPK * T * 2; }
__host__ __device__ inline int regen_off_table(int Kls, int T) { return regen_off_mins(Kls, T) + 4; }
__host__ __device__ inline int regen_record_length(int Kls, int T) { return (regen_off_table(Kls, T) + LAD_N * 3 + 3) / 4 * 4; }

// ---- launchers (defined in the .hip files) ---------------------------------------------
bool launch_rollout_point(const RolloutArgs& a, const PointScene& sc, hipStream_t s);
void launch_rollout_point_nav(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s);
void launch_rollout_point_push(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s);
void launch_rollout_point_pull(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s);
void launch_rollout_point_pushpull(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s);
void launch_transpose_noise(const float* src_ktn, float* dst_tkn, int K, int T, int nu,
                            hipStream_t s);
void launch_spline_noise(const float* knots, float* noise, int Kl, int nu, int n_knots, int T, int degree,
                         double smoothing, hipStream_t s);
void launch_sample_noise(const RolloutArgs& a, float* out, hipStream_t s);
void launch_halton_knots(float* knots, int k0, int n, int ncol, const int* primes_dev, const int* perm_dev,
                         const int* perm_off_dev, hipStream_t s);
struct OrderScene {   // where the scene's objects are when the wavefront order is computed
    const float* sim_root;  // != null: box / dyn-obs positions are read from the bound root_state tensor
    int sim_box, sim_dyn;
    float bx, by, dx, dy;   // otherwise these (host world)
    float ox, oy;           // obstacle
};
size_t wave_order_temp_bytes(int Kl);
hipError_t launch_wave_order(const float* noise, int Kl, int T, int nu, float s0, float s1, int half_local,
                             const OrderScene& os, void* scratch, size_t temp_bytes, int* order, float* noise_sorted,
                             const int* specials /* null, or 3 local indices (-1: none) */, hipStream_t s);
void launch_gather_rows(const float* src, const int* order, float* dst, int Kl, int rows, hipStream_t s);
void launch_weights(const UpdateArgs& a, hipStream_t s);
void launch_wsum(const UpdateArgs& a, hipStream_t s);
void launch_update_small(const UpdateArgs& a, hipStream_t s);
bool update_small_applies(const UpdateArgs& a);
void launch_finalize(const UpdateArgs& a, hipStream_t s);
void launch_mix(const UpdateArgs& a, hipStream_t s);
void launch_cov_update(const float* actions, const float* w, const float* mean, float* part, float* cov, int K, int T,
                       int nu, hipStream_t s);
void launch_local_topk(const UpdateArgs& a, hipStream_t s);
void launch_regen_fast(const UpdateArgs& a, hipStream_t s);
void launch_p3_search(const UpdateArgs& a, hipStream_t s);          // mixed-table search (+ top-k merge) of shard_mix = 3
void launch_p3_local_weights(const UpdateArgs& a, hipStream_t s);   // k_apply_weights<true> over the local costs
void launch_p3_done(const UpdateArgs& a, hipStream_t s);            // after the second exchange: sums, best rows, plan
int init_ladder_table();   // update.hip: the beta ladder into constant memory (per device context)
int regen_chunks(int Kg);   // workgroups per time step of k_regen_part
int rollout_lanes_for(int Kl);
// MI355X: 256 CUs x 4 SIMDs.  A rollout launch with more wavefronts than that is throughput-bound: the point_env
// kernels then run in their two-waves-per-SIMD build (rollout_point_kernel.hpp)
constexpr int M3_SIMDS = 1024;
inline bool rollout_two_waves(int wavefronts) { return wavefronts > M3_SIMDS; }
inline bool rollout_three_waves(int wavefronts) { return wavefronts > 4 * M3_SIMDS; }   // (measured: equal at 4 per SIMD, -11 % at 8)
int mins_workgroups(int Kg);
int topk_workgroups(int Kg);
int ladder_workgroups(int Kg);
int wsum_chunks(int Kl);
int apply_workgroups(int Kg);
void launch_mins(const UpdateArgs& a, hipStream_t s);
void launch_ladder(const UpdateArgs& a, hipStream_t s);
void launch_ladder_search(const UpdateArgs& a, hipStream_t s);
void launch_fused_large(const UpdateArgs& a, hipStream_t s);

// p2p.hip: device-side exchange of the records over peer-mapped memory
struct P2PArgs {
    const float* rec;                      // this rank's record
    int rec_len, rec_stride, n_ranks, rank, slot, seq;   // rec_stride: floats between two slots of a block (rec_len rounded up to 4)
    int plain_memory;                      // the block is ordinary (cached) device memory: system-scope fences needed
    unsigned long long timeout_ticks;      // of the 100 MHz wall clock
    int* err;                              // device word in the own block: 0, or 1 + the rank that never arrived
    float* peer_data[MIX_MAX_RANKS];       // data part of every rank's exchange block: [2][n_ranks][rec_stride]
    int* peer_flags[MIX_MAX_RANKS];        // flag part: [2][MIX_MAX_RANKS]
};
constexpr size_t P2P_HDR_BYTES = 1024;    // flags [2][32] ints + error word, then the data part
void launch_p2p_put(const P2PArgs& a, hipStream_t s);
void launch_p2p_wait(const P2PArgs& a, hipStream_t s);
void launch_p2p_exchange(const P2PArgs& a, hipStream_t s);   // both in one launch

// step mode
struct SimViews {
    float* dof_state;          // [Kl][2*ndof]
    float* root_state;         // [Kl][nA][13]
    float* rigid_body_state;   // [Kl][nB][13]
    float* net_contact_force;  // [Kl][nB][3]
    int n_actors, n_bodies;
    int box_actor, dyn_actor, robot_actor;  // panda_env: box = cubeA, dyn = cubeB
    int box_body, dyn_body, robot_body;     // point: robot_body = link_y (last body);
                                            // panda: robot_body = panda_link0 (first of 11)
    int table_body, shelf_body;             // panda_env only
    int obs_actor, obs_body;                // panda_env only: the dyn-obs plate
};
void launch_sim_pull(const SimViews& v, float* world /*[NW][Kl]*/, int Kl, hipStream_t s);
void launch_sim_shift_pull(const SimViews& v, float* world, int Kl, int actor, float dx, float dy, float dz, hipStream_t s);
void launch_sim_push(const SimViews& v, const float* world, int Kl, hipStream_t s);
void launch_sim_step(const PointScene& sc, const SimViews& v, float* world, const float* u /*[Kl][2]*/, float* u_keep,
                     int Kl, hipStream_t s);   // step + refresh of the views
void launch_sim_forces(const SimViews& v, float* world, const float* f /*[Kl][nB][3]*/, int Kl,
                       hipStream_t s);
void launch_sim_cost(const CostParams& cp, float* world, int Kl, int k0, float* cost,
                     hipStream_t s);
void launch_sim_suction(const SimViews& v, float* world, int Kl, float kp, float thresh, float reach,
                        const float* action, int apply, float* forces, int* flags, const int* gate, hipStream_t s);

constexpr int NW = 28;  // floats per env in the step-mode SoA world (PointWorld fields)
constexpr int NWP = 77; // same for the panda_env (PandaWorld fields, rollout_panda.hip)

// panda_env
constexpr int REACH_REC = 17;   // floats the reach cost reads of a (step, sample): rollout_panda.hip
struct PandaArgs {
    float world0[57];  // q9 qd9 | cubeA13 | cubeB13 | dyn-obs13 (pos3 quat4 vel3 angvel3)
    int cubeA_actor, cubeB_actor, obs_actor;
    PandaCostParams cp;
    int shadows;       // the last (two) sample slot(s) of every wavefront re-simulate sample 0 (and K / 2): quirk Q8
    int lps;           // lanes per sample of the rollout kernel: 0 = by size, 1, 8, 16 (m3_set_panda_lanes_per_sample)
    float* reach_rec;            // [T][REACH_REC][Kl] or null: the reach cost is formed after the rollout (k_panda_reach_cost), no shadow slots
    int* busy_hint;              // device address of the handle's hint word (host memory, mapped): 1 + the share, in 1/1000, of the
                                 // launch's (sample, substep) pairs with the gripper within reach of a box; written by the last wavefront
    unsigned long long* busy_count;   // device scratch of that: bits 0-23 wavefronts finished, bits 24-63 the sum so far
    int reach_busy;              // the host's reading of the last reports, with hysteresis: the reach command runs with 8 lanes per sample
};
int launch_rollout_panda(const RolloutArgs& a, const PandaArgs& pa, const PandaScene& sc, hipStream_t s,
                         int* lps_used = nullptr);   // returns its workgroups; *lps_used = the kernel form it chose
void launch_psim_step(const PandaScene& sc, const SimViews& v, float* world, const float* u, float* u_keep, int Kl,
                      hipStream_t s);
void launch_psim_pull(const PandaScene& sc, const SimViews& v, float* world, int Kl, hipStream_t s);
void launch_psim_push(const PandaScene& sc, const SimViews& v, const float* world, int Kl, hipStream_t s);
void launch_psim_cost(const PandaScene& sc, const PandaCostParams& cp, const float* world, int Kl, int k0, bool env0_cube,
                      float* cost, hipStream_t s);

}  // namespace m3

struct m3_handle {
    m3_config cfg;
    int* order = nullptr;          // [Kl] lane slot -> local sample (null: identity), sampler.hip
    void* order_scratch = nullptr;
    float* noise_sorted = nullptr; // [T][Kl][nu] noise rows in wavefront order
    size_t order_temp_bytes = 0;
    bool wave_order = true;        // m3_set_wave_order
    bool order_valid = false;
    bool order_dirty = true;       // recomputed by the next m3_rollout (needs the world)
    unsigned last_noise_call = 0;  // h->calls at the last m3_set_noise*
    int noise_churn = 0;           // consecutive noise uploads fewer than 16 commands apart
    bool relabel_pending = false;  // m3_relabel_samples: done by the next m3_rollout
    bool relabelled = false;       // the noise rows ARE in wavefront order (identity order from then on)
    m3::PointScene scene;
    m3::PandaScene pscene;
    float pworld0[57];
    hipStream_t stream = nullptr;
    std::string err;
    // objective
    int task = M3_TASK_PUSH;
    float goal[7] = {0, 0, 0, 0, 0, 0, 1};
    int gripper_cmd = 0;
    int avoid_dyn_obs = 0;             // m3_set_avoid_dyn_obs (extension; 0 = the reference's compute_cost)
    // world
    float world0[18];
    const float* world0_bound = nullptr;  // device, 18 floats (filled by world_from_sim)
    const float* bind_dof = nullptr;
    const float* bind_root = nullptr;
    int bind_nact = 0, bind_box = 0, bind_dyn = 0, bind_obs = 0;
    bool have_noise = false;
    bool cov_active = false;       // cfg.update_cov on a single-mode halton-spline planner (mppi.py:508-516)
    float* noise_mats = nullptr;   // device [2][nu][nu]: chol(noise_sigma) | noise_sigma^-1 (cfg.full_sigma)
    bool regen = false;            // one-collective multi-modal sharding (UpdateArgs::regen)
    bool regen_fast = false;       // ... with per-rank ladder tables in the records (cfg.shard_mix >= 2)
    bool p3 = false;               // cfg.shard_mix == 3: O(K_local) work after the gather, second small exchange
    const float* recb_src = nullptr;   // second records as exchanged by m3_p2p_exchange(h, 1) (else M3_BUF_RECORDS_B_ALL)
    int recb_stride = 0;
    float* noise_all = nullptr;    // regen: [n_ranks][T][Kl][nu]; buf[M3_BUF_NOISE] aliases this rank's block
    int* local_top_idx = nullptr;  // regen: top_idx of the local pre-gather selection (scratch)
    unsigned calls = 0;
    int lanes_override = 0;  // 0 = automatic (rollout_lanes_for)
    int* panda_busy_hint = nullptr;   // hipHostMalloc (m3_create, panda_env): see PandaArgs::busy_hint
    int* panda_busy_hint_dev = nullptr;   // the same word as the device sees it
    unsigned long long* panda_busy_count = nullptr;
    float* panda_reach_rec = nullptr;     // PandaArgs::reach_rec (m3_create: unsharded panda handles with K <= PANDA_REACH_REC_MAX_K)
    bool panda_reach_deferred = true;     // m3_set_panda_reach_cost_kernel
    int panda_reach_busy = 0;
    int panda_lps_used = 0;           // the form of the last panda rollout (m3_panda_lanes_per_sample_used)
    int panda_lps = 0;       // 0 = automatic (rollout_panda.hip: panda_lps_for), 1, 16
    // device buffers
    void* buf[M3_BUF_COUNT] = {};
    long long nbytes[M3_BUF_COUNT] = {};
    float* world0_dev = nullptr;
    float* action_out = nullptr;  // caller-owned destination of the plan (m3_set_action_out)
    m3::VI* topk_cand = nullptr;
    float* part_min = nullptr;
    float* lad = nullptr;
    float* wpart = nullptr;
    float* apart = nullptr;  // + 16 floats of SearchOut at the front
    int* wcount = nullptr;
    // unsharded multi-modal update with K > 8192 in three launches (update.hip: k_ladder_search)
    float* wave_min = nullptr;     // [rollout workgroups][3] minima left by the rollout (wave_min.hpp)
    int wave_min_rows = 0;         // rows the last rollout wrote
    bool use_wave_min = false;     // inside m3_command: the costs are the rollout's
    int* lflag = nullptr;
    int lad_epoch = 0;
    bool five_launches = false;    // m3_set_update_launches(h, 5)
    int ladder_spins = 1 << 18;    // bounded wait of the in-launch ladder exchange (~20 ms); m3_set_ladder_spins
    float* sim_world = nullptr;  // step mode SoA [NW][Kl]
    float* sim_u = nullptr;      // [Kl][nu]
    float* noise_stage = nullptr;
    m3::SimViews views{};
    bool views_bound = false;
    // device-side exchange (p2p.hip)
    void* xb = nullptr;                 // own exchange block: header (flags, error word) + [2][n_ranks][rec_len]
    size_t xb_bytes = 0;
    int xb_kind = 0;                    // 1 uncached, 2 fine-grained, 3 plain device memory
    int xb_first_kind = 1;              // where the allocation's fallback chain starts (m3_p2p_set_memory_kind)
    void* peer_base[m3::MIX_MAX_RANKS] = {};   // every rank's block as mapped here (own: xb)
    bool peer_ipc[m3::MIX_MAX_RANKS] = {};     // opened with hipIpcOpenMemHandle (closed in m3_destroy)
    bool p2p_ready = false;
    int p2p_seq[2] = {0, 0};           // per channel (0: records, 1: second records of shard_mix = 3)
    int p2p_first_ms = 30000, p2p_ms = 500;   // wait time-outs (m3_p2p_set_timeout_ms)
    int records_stride = 0;             // floats between two records of records_src
    const float* records_src = nullptr; // != null: the records of the exchange just enqueued (consumed by m3_finalize)
    // timing
    bool timing = false;
    hipEvent_t ev[4] = {};
};
