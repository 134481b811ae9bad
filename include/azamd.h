/*
 * azamd.h -- C ABI of libazamd.so, the MI355X (gfx950) engine for the go_zero self-play hot path:
 * Go rules + feature planes, batched PUCT search over device-resident trees for G lockstep games,
 * and policy/value network inference.
 *
 * The upstream project is pure Python and has no FFI of its own; each entry point below names the
 * upstream Python function (file:line under go_zero/) whose work it takes over. A Python binding
 * (ctypes) that re-creates the upstream call surface on top of these functions ships in
 * alphazero_amd/ and go_zero/.
 *
 * Conventions
 *   - every function returns 0 on success or a negative az_status; az_last_error() gives the text
 *     for the calling thread;
 *   - pointers named *_dev are device pointers owned by the caller, never freed or retained;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously unless the
 *     function is documented as synchronising;
 *   - actions are r*n+c, pass = n*n; colours are BLACK=+1, WHITE=-1, EMPTY=0 (go.py:10-29);
 *   - an engine is not thread-safe; use one engine per host thread / stream.
 */
#ifndef AZAMD_H
#define AZAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    AZ_OK = 0,
    AZ_ERR_ARG = -1,      /* bad argument / unsupported size */
    AZ_ERR_HIP = -2,      /* a HIP runtime call failed */
    AZ_ERR_STATE = -3,    /* call made in the wrong phase of a search */
    AZ_ERR_ILLEGAL = -4,  /* illegal move (go.py:179-180 ValueError) */
    AZ_ERR_NOCHILD = -5,  /* "No child selected in MCTS" (mcts.py:114-115 RuntimeError) */
    AZ_ERR_CAPACITY = -6  /* a per-game tree ran out of node slots */
} az_status;

typedef struct az_engine az_engine;
typedef struct az_net az_net;

/* Engine configuration. Mirrors GoState.new_game (go.py:62) and MCTS.__init__ (mcts.py:31-57). */
typedef struct {
    int32_t board_size;       /* n, 2..19 */
    int32_t history_len;      /* H, 1..16 */
    int32_t games;            /* G lockstep games held on the device */
    int32_t simulations;      /* num_simulations */
    int32_t inference_batch;  /* inference_batch_size (clamped to >= 1) */
    int32_t topk_expand;      /* topk_expand (clamped to >= 0; 0 = expand every legal action) */
    int32_t dedup_leaves;     /* 1: a leaf reached by several descents of one batch is evaluated once
                                 (results are identical either way: the evaluator is a pure function) */
    int32_t device;           /* HIP device ordinal */
    double komi;
    double c_base, c_init;    /* PUCT constants */
    double dirichlet_eps;     /* root noise weight */
    double virtual_loss;      /* clamped to >= 0 */
} az_config;

const char* az_last_error(void);
/* 64-bit Zobrist key used for the situational-superko set; index = 2*point + (colour==WHITE) for
 * stones, 2*361 + (to_play==WHITE) for the side to move. Same role as go.py:53-59 _state_hash. */
uint64_t az_zobrist_key(int32_t index);

int az_engine_create(const az_config* cfg, az_engine** out);
int az_engine_destroy(az_engine* e);
/* sizes a caller needs to allocate buffers */
int az_engine_action_size(const az_engine* e);    /* A = n*n+1           (go.py:15-16)  */
int az_engine_feature_planes(const az_engine* e); /* C = 2H+5            (go.py:32-34)  */
int az_engine_seen_capacity(const az_engine* e);  /* entries per game in export/import  */
int az_engine_max_rows(const az_engine* e);       /* upper bound on pending rows        */

/* ---- rules on the G root states --------------------------------------------------------- */
/* GoState.new_game (go.py:61-74) for the games with reset_mask_dev[g] != 0 (NULL = all). */
int az_rules_reset(az_engine* e, const uint8_t* reset_mask_dev, void* stream);
/* GoState.play (go.py:159-203) with actions_dev[g] (-1 = leave game g untouched). status_dev[g]
 * receives 0, or AZ_ERR_ILLEGAL and the state is left unchanged. */
int az_rules_play(az_engine* e, const int32_t* actions_dev, int32_t* status_dev, void* stream);
/* legal_actions as a mask uint8 [G, A] (go.py:145-157; pass is always 1). */
int az_rules_legal_mask(az_engine* e, uint8_t* mask_dev, void* stream);
/* to_tensor (go.py:265-304). layout AZ_FEAT_F32_NCHW: float32 [G, C, n, n];
 * AZ_FEAT_BF16_NHWC: bf16 [G, n*n, 32] (channels padded with zeros). */
enum { AZ_FEAT_F32_NCHW = 0, AZ_FEAT_BF16_NHWC = 1 };
int az_rules_features(az_engine* e, void* out_dev, int32_t layout, void* stream);
/* score/winner/is_terminal (go.py:205-263): score_dev double [G,2] (black, white incl. komi),
 * info_dev int32 [G,8] = to_play, consecutive_passes, move_count, black_captures, white_captures,
 * last_move (-1 none), is_terminal, winner. Either pointer may be NULL. */
int az_rules_status(az_engine* e, double* score_dev, int32_t* info_dev, void* stream);
/* Whole-state export/import (device buffers): board int8 [G,n*n]; history int8 [G,H,n*n] newest
 * first; scalars int32 [G,8] = to_play, passes, move_count, black_caps, white_caps, last_move,
 * history_count, seen_count; seen uint64 [G, seen_capacity]. */
int az_rules_export(az_engine* e, int8_t* board_dev, int8_t* hist_dev, int32_t* scalars_dev, uint64_t* seen_dev,
                    void* stream);
int az_rules_import(az_engine* e, const int8_t* board_dev, const int8_t* hist_dev, const int32_t* scalars_dev,
                    const uint64_t* seen_dev, void* stream);

/* ---- batched PUCT search (mcts.py:165-225 _run_single_root_search for every game) -------- */
/* Start a search from the current root states. Games whose root is terminal, or with
 * enable_mask_dev[g] == 0, sit the search out (mcts.py:234). Pending rows = the active roots. */
int az_mcts_begin(az_engine* e, const uint8_t* enable_mask_dev, void* stream);
/* Select phase of one batch (mcts.py:181-199): min(B, remaining) descents per game with virtual
 * loss; pending rows = leaves to evaluate. */
int az_mcts_select(az_engine* e, void* stream);
/* Number of pending rows. Synchronises `stream`. Also reports and clears per-game failures
 * (returns AZ_ERR_NOCHILD / AZ_ERR_CAPACITY). */
int az_mcts_pending(az_engine* e, int32_t* rows_out, void* stream);
/* Root-evaluation reuse (enable != 0), for evaluators that are pure functions of a position independent of the
 * batch it is in (az_net_forward is; a framework model on the host need not be): the state a game is in after
 * az_rules_play was evaluated as a child of the previous root, so its logits are carried over and the next
 * az_mcts_begin asks for no evaluation of that root. az_mcts_pending then counts only the rows to evaluate;
 * az_mcts_gather still fills az_mcts_gather_rows rows (the carried-over roots last: their legal masks are needed),
 * and az_mcts_expand takes logits for the first az_mcts_pending rows (NULL if there are none). Off by default.
 * Use one evaluator per engine while it is on. Synchronises the device. */
int az_mcts_set_root_reuse(az_engine* e, int32_t enable);
int az_mcts_gather_rows(az_engine* e, int32_t* rows_out, void* stream);
/* Asynchronous mode (enable != 0): az_mcts_gather and az_mcts_expand never synchronise; launches are
 * sized by upper bounds (G rows for the roots, az_engine_max_rows for a batch) and surplus workers
 * exit on the device-side row count, so a whole search can be enqueued without host round trips.
 * Failures are then reported by the next az_mcts_pending / az_mcts_stats call. */
int az_mcts_set_async(az_engine* e, int32_t enable);
/* All remaining batch rounds of a search in one call (the loop of mcts.py:181-220 once the root is expanded): select,
 * leaf gather into `feat_dev`, az_net_forward_indirect of `net` into `logits_dev` / `values_dev`, expand + backup, until
 * the simulations are used up. Needs asynchronous mode and buffers for az_engine_max_rows rows. use_graph != 0: the
 * launches are captured into a HIP graph the first time and replayed by later calls with the same network and buffers
 * (a move's rounds always launch with the same arguments, the row counts live on the device); taken only when
 * az_net_capturable(net, rows) holds and the stream can be captured (not the legacy default stream), otherwise the
 * rounds are enqueued one by one. Returns 1 when the graph was launched, 0 when the rounds were enqueued, < 0 on error.
 * Failures of the rounds themselves surface at az_mcts_stats, as in asynchronous mode. A captured graph refers to the
 * network's weights and scratch buffers: keep the network alive as long as the engine may replay it (az_engine_destroy and
 * az_mcts_set_root_reuse drop the graph). az_net_load, az_net_reserve, az_net_set_tower_workgroups and
 * az_net_set_device_choice on that network make the next call capture anew instead of replaying. */
int az_mcts_run_rounds(az_engine* e, az_net* net, void* feat_dev, int32_t layout, float* logits_dev, float* values_dev,
                       int32_t use_graph, void* stream);
/* Device address of the int32 pending-row count (valid on `stream` after az_mcts_begin/az_mcts_select),
 * for consumers that size their work on the device (az_net_forward_indirect). */
int az_mcts_rows_device(az_engine* e, const int32_t** rows_dev_out);
/* Leaf gather: feature planes of the pending rows, dense, in row order (mcts.py:61). */
int az_mcts_gather(az_engine* e, void* feat_out_dev, int32_t layout, void* stream);
/* Row bookkeeping for callers that want it: game index int32 [rows] of each pending row. */
int az_mcts_row_games(az_engine* e, int32_t* row_game_dev, void* stream);
/* Expand + backup (mcts.py:66-91 masked softmax in float64, :118-163 expand with top-k, :205-218
 * backup). logits_dev float32 [rows, A], values_dev float32 [rows]. After az_mcts_begin this expands
 * the roots: noise_dev (double [G, A], Dirichlet sample of game g in its first n_legal entries, or
 * NULL) is mixed into the root priors (mcts.py:121-124). Returns 1 when all simulations are done. */
int az_mcts_expand(az_engine* e, const float* logits_dev, const float* values_dev, const double* noise_dev,
                   void* stream);
/* Number of legal actions of every root (host needs it to draw the Dirichlet sample). int32 [G]. */
int az_mcts_root_legal_counts(az_engine* e, int32_t* counts_dev, void* stream);
/* visits int32 [G, A] and root value double [G] (mcts.py:222-225). Inactive games give zeros. */
int az_mcts_result(az_engine* e, int32_t* visits_dev, double* root_value_dev, void* stream);
/* Priors of the pending rows' nodes after the last az_mcts_expand (double [rows, A], 0 where no
 * child) -- for parity tests of the masked softmax. */
int az_mcts_last_priors(az_engine* e, double* priors_dev, void* stream);
/* Counters since az_mcts_begin: int64 [8] = simulations, rows evaluated, rows before dedup,
 * sum of leaf depths, max depth, nodes allocated (max over games), terminal leaves, batches. Synchronises. */
int az_mcts_stats(az_engine* e, int64_t* out8_host, void* stream);

/* ---- policy/value network (model.py:43-97 AlphaZeroNet.forward, eval mode) ---------------- */
typedef struct {
    int32_t board_size, in_planes, channels, blocks, use_se, se_hidden, value_hidden;
    int32_t precision; /* AZ_NET_BF16 or AZ_NET_F32 */
    int32_t device;
} az_net_config;
enum { AZ_NET_BF16 = 0, AZ_NET_F32 = 1 };

int az_net_create(const az_net_config* cfg, az_net** out);
int az_net_destroy(az_net* net);
/* Number of float32 words in the flat parameter blob and the offset of a named tensor in it. The
 * blob is the state_dict (model.py:61-87) flattened in the order documented in az_net_param_layout. */
int64_t az_net_param_count(const az_net* net);
/* Upload weights: params_host is the flat float32 blob (host memory). BatchNorm running statistics
 * are folded into the preceding convolution. Synchronises. */
int az_net_load(az_net* net, const float* params_host, int64_t count);
/* Forward pass for `rows` positions. feat_dev layout must match the precision (bf16 NHWC for
 * AZ_NET_BF16, float32 NCHW for AZ_NET_F32). logits_dev float32 [rows, A], values_dev float32 [rows]. */
int az_net_forward(az_net* net, const void* feat_dev, int32_t rows, float* logits_dev, float* values_dev,
                   void* stream);

/* Allocate the activation scratch for forwards of up to `rows` positions now, so that az_net_forward /
 * az_net_forward_indirect calls of that size allocate nothing and never wait for the device (without it the scratch
 * grows at the first larger call; a replaced buffer is kept until az_net_destroy because enqueued launches may
 * still use it). A layer-by-layer network (boards above 9x9, 256 channels) also grows its activation arrays to `rows`
 * boards, up to 2 GiB per array, so that az_net_forward_indirect covers an upper bound of `rows` in one pass. Same role
 * as the lazily built torch workspace behind model(x) in mcts.py:59-64. */
int az_net_reserve(az_net* net, int32_t rows);

/* *fused = 1 if the bf16 network runs as the single fused tower kernel (boards up to 9x9 with 32, 64, 128 or 192 channels;
 * 11x11 to 16x16 with 128 or 192 channels; 17x17 to 19x19 with 128 channels), 0 if it runs layer by layer (the other shapes,
 * every 256-channel tower) or in float32. */
int az_net_is_fused(az_net* net, int32_t* fused);
/* az_net_forward with the row count read on the device from *rows_dev (<= max_rows): no host round trip. bf16
 * networks of either kind; a layer-by-layer network runs whole chunks over max_rows whose surplus workgroups exit. */
int az_net_forward_indirect(az_net* net, const void* feat_dev, const int32_t* rows_dev, int32_t max_rows,
                            float* logits_dev, float* values_dev, void* stream);
/* az_net_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no counterpart upstream. Fused bf16 networks on boards up to 9x9 only (az_net_static_max_rows is 0
 * for the others). The activation
 * scratch is a 64 MiB device array inside the library (one copy per GPU, held until the process ends): residual
 * stream for 256 workgroups, then the head activations, which bounds `rows` by az_net_static_max_rows (2 696 rows
 * for 9x9 / 192 channels); more rows return AZ_ERR_CAPACITY. The call allocates and frees nothing and records no
 * events. Every network and every call on a GPU shares that GPU's array: calls that overlap on one GPU (other
 * streams or threads) race; calls on one stream are ordered and safe. Results equal az_net_forward's bit for bit. */
int az_net_forward_static(az_net* net, const void* feat_dev, int32_t rows, float* logits_dev, float* values_dev,
                          void* stream);
int64_t az_net_static_max_rows(const az_net* net);

/* How many launch sequences az_net_forward_indirect makes for an upper bound of max_rows: 1 for a fused network; a
 * layer-by-layer network runs one sequence per capacity of its activation arrays (az_net_reserve(max_rows) makes that 1
 * unless an array would pass 2 GiB), and the sequences past the live rows are empty launches that still cost their
 * dispatch, so a caller may prefer one host round trip per batch when this exceeds 1. */
int az_net_indirect_passes(const az_net* net, int32_t max_rows);

/* 1 if az_net_forward_indirect over max_rows rows is launches only (one pass, scratch reserved by az_net_reserve, launch
 * timing off) and may therefore be recorded by a stream capture; 0 otherwise. No upstream counterpart. */
int az_net_capturable(const az_net* net, int32_t max_rows);

/* Timing of the dominant kernel (the fused tower) with HIP events recorded on the launch stream.
 * If the three output pointers are non-NULL they receive the total milliseconds, launch count and
 * row count accumulated since profiling was last enabled (synchronises on the recorded events);
 * then profiling is switched on (enable != 0) or off with the counters cleared. bf16 path only.
 * No upstream counterpart: measurement aid for bench.py. */
int az_net_profile(az_net* net, int32_t enable, double* tower_ms, int64_t* launches, int64_t* rows);
/* [start, end] in ms of each timed launch since profiling was enabled (before az_net_profile reads and
 * clears them), on one time line per device that all networks and streams share, so that launches
 * overlapping on different streams can be merged. Writes min(count, capacity) pairs. */
int az_net_profile_spans(az_net* net, double* spans_ms, int64_t capacity, int64_t* count);
/* Upper limit of workgroups per launch of the fused residual-tower kernel, in units of one 512-thread workgroup per
 * compute unit (default: the number of compute units; the 256-thread kernels, of which a compute unit holds two, launch
 * twice the limit), each looping over its share of the boards. A larger limit gives shorter-lived workgroups, which lets
 * launches on other streams interleave at board-group granularity but costs more than it gains from two streams on
 * (4096 games of 9x9 192ch x 12 in two lanes: 16.2 M simulations/s with the default, 15.5 M with 1024). Synchronises the
 * device. */
int az_net_set_tower_workgroups(az_net* net, int32_t workgroups);
/* az_net_forward_indirect under a row bound of up to 32 times the low-latency tower variant's limit launches that
 * variant and the main kernel, and the count in device memory picks the one that works (on != 0, the default).
 * The low-latency variant holds every compute unit with one workgroup, which is what a single stream of searches
 * wants and what three or more streams sharing the device do not: their driver switches the choice off (on == 0)
 * and the main kernel runs whatever the count. Results are the same bits either way. bf16 path only; no upstream
 * counterpart (host-side switch of the GPU engine). */
int az_net_set_device_choice(az_net* net, int32_t on);

/* ---- host-side random streams (no device work) ------------------------------------------------ */
/* One legacy-NumPy-compatible MT19937 stream per game. The upstream code draws from the global
 * np.random state: np.random.dirichlet([alpha]*k) for the root noise (mcts.py:122) and the uniform
 * behind np.random.choice(len(pi), p=pi) (mcts.py:281). With one stream per game, game g in a
 * lockstep batch draws what a sequential upstream run seeded with np.random.seed(seed_g) draws. */
typedef struct az_rng az_rng;
int az_rng_create(int32_t count, az_rng** out);
int az_rng_destroy(az_rng* r);
int az_rng_seed(az_rng* r, int32_t index, uint32_t seed);
/* out (host, double [count, stride]) gets dirichlet([alpha]*counts[g]) in its first counts[g] entries. */
int az_rng_dirichlet(az_rng* r, const int32_t* counts_host, double alpha, double* out_host, int32_t stride,
                     int32_t threads);
/* One uniform double in [0,1) per stream with mask_host[g] != 0 (NULL = all): RandomState.random_sample(). */
int az_rng_random_sample(az_rng* r, const uint8_t* mask_host, double* out_host);

/* ---- self-play lane: a whole move of every lockstep game per enqueue ----------------------------
 * Takes over, for the G games of an engine, one iteration of the loop of selfplay_game (go_zero/train.py:322-343)
 * around MCTS.run (mcts.py:227-284): root noise, search, visits -> policy target with the move's temperature
 * (mcts.py:270-279), the sampled action (mcts.py:281), difficulty (train.py:336-337), resignation (train.py:339-341),
 * the training sample's planes (go.py:265-304), GoState.play (train.py:342) and the terminal test / winner
 * (train.py:322, :345). Everything but the random draws runs on the device; the host waits once per move.
 * The engine must be in asynchronous mode and `net` a bf16 network with az_net_reserve(az_engine_max_rows) done. */
typedef struct az_lane az_lane;
typedef struct {
    int32_t temp_moves, temp_mid_moves;       /* train.py:323-328: move index below which temp_start / temp_mid apply */
    double temp_start, temp_mid, temp_end;
    double resign_threshold;                  /* train.py:339 */
    int32_t disable_resign, resign_min_move;  /* upstream resigns from move index 12 on */
    double dirichlet_alpha;                   /* mcts.py:122 */
    int32_t collect_samples;                  /* keep (planes, pi, to_play, difficulty) of every searched position */
    int32_t rng_threads;                      /* host threads drawing the Dirichlet samples */
    int32_t use_graph;                        /* replay the batch rounds from a HIP graph where the network allows it */
} az_lane_config;
/* What az_lane_finish reports. Array pointers are host memory owned by the caller, G entries (rows) each, NULL to skip. */
typedef struct {
    int64_t counters[8];   /* out: simulations, rows evaluated (root rows included), rows before dedup, sum of leaf depths,
                              max depth, max nodes, terminal leaves, positions searched */
    double host_seconds_rng; /* out: host time spent drawing this move's random numbers */
    int32_t n_over;        /* out: games that ended with this move (resignation, two passes, move cap) */
    int32_t* over_slot;    /* out [n_over]: their slots, ascending; the slot already holds a fresh game */
    int32_t* over_winner;  /* out [n_over]: +1 black, -1 white, 0 draw (train.py:345) */
    uint8_t* bits;         /* out [G][row_bytes], row_bytes = ceil((2H+3)*n*n / 8): binary planes of the searched state,
                              numpy.packbits order (collect_samples) */
    float* caps;           /* out [G][2]: values of the two capture planes (collect_samples) */
    float* pi32;           /* out [G][A]: policy target (collect_samples) */
    int32_t* to_play;      /* out [G]: side to move in the searched state */
    double* difficulty;    /* out [G]: max(0.01, 1 - max pi) */
    int32_t* actions;      /* out [G]: the sampled action (-1: empty slot) */
    double* root_value;    /* out [G] */
} az_lane_result;
/* pow_tables_host: double [3][pow_len], pow_len > simulations: v ** (1 / T) for v = 0..pow_len-1 and T = temp_start,
 * temp_mid, temp_end, computed by the caller with the pow whose bits it wants (upstream: numpy.power, mcts.py:274).
 * feat_dev / logits_dev / values_dev: device buffers for az_engine_max_rows rows, as for az_mcts_run_rounds. `rng`: one
 * stream per slot; the caller seeds a slot's stream when it deals a game to it. */
int az_lane_create(az_engine* e, az_net* net, az_rng* rng, const az_lane_config* cfg, const double* pow_tables_host,
                   int32_t pow_len, void* feat_dev, int32_t layout, float* logits_dev, float* values_dev, az_lane** out);
int az_lane_destroy(az_lane* lane);
/* Host arrays the caller edits between moves: slot_on uint8 [G] (1: the slot holds a running game), move_idx int32 [G]
 * (moves played so far in the slot's game; az_lane_finish advances it and zeroes it when a game ends). */
int az_lane_slots(az_lane* lane, uint8_t** slot_on_host, int32_t** move_idx_host);
/* Evaluate the roots of the current states (az_mcts_begin + root rows through the network) without waiting: after
 * az_rules_reset / az_rules_play from outside the lane (new lane, opening-book moves) and before az_lane_launch. */
int az_lane_prepare(az_lane* lane, void* stream);
/* Enqueue one move of every running game: draws each game's Dirichlet sample and move-choice uniform from its stream,
 * then root expansion, all batch rounds, the move itself and the evaluation of the next roots. refill != 0: a slot whose
 * game ends with this move starts a fresh game at once (the caller deals it an id after az_lane_finish, or clears its
 * slot_on flag); refill == 0: such a slot goes idle. Returns 1 when the rounds were replayed from a graph, 0 otherwise,
 * < 0 on error. */
int az_lane_launch(az_lane* lane, int32_t refill, void* stream);
/* Wait for the move enqueued by az_lane_launch and report it. Failures of the search (AZ_ERR_NOCHILD, AZ_ERR_CAPACITY) and
 * an illegal sampled move (AZ_ERR_ILLEGAL, "Illegal move: (r, c)") surface here. */
int az_lane_finish(az_lane* lane, az_lane_result* result);

/* Do kernels enqueued on streams `a` and `b` run at the same time? HIP maps streams onto a few hardware queues
 * (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one run their kernels one after the other: lanes on such a
 * pair lose their overlap (two lanes of 256 games of a 64ch x 4 network: 2.9 instead of 1.9 ms per move). Launches a
 * one-wave kernel that waits `microseconds` (clamped to 20..2000) on each stream and compares the wall time with one wait:
 * *overlap = 1 when the two ran concurrently. No counterpart upstream (its self-play workers are processes,
 * train.py:395-463); alphazero_amd/selfplay.py picks its lanes' streams with it. Synchronises both streams. */
int az_streams_overlap(void* stream_a, void* stream_b, int32_t microseconds, int32_t device, int32_t* overlap);

/* ---- replay batch assembly (go_zero/train.py:147-176 ReplayBuffer.sample, :187-207 symmetries) ----
 * Gathers `count` sampled positions idx[i] from a device-resident replay store and applies dihedral
 * symmetry sym[i] (0..7, the numbering of train.py:187-192) to planes and policy. Store layout: `bits`
 * [capacity][row_bytes] = the 2H+3 binary planes of to_tensor() packed row-major, most significant bit
 * first; `caps` [capacity][2] = the two capture planes' values; `pi` [capacity][A]. `perm` [8][n*n]:
 * source point of every destination point under each symmetry. Outputs float32 [count][2H+5][n][n] and
 * [count][A], exactly what ReplayBuffer.sample stacks on the host. */
int az_replay_gather(const uint8_t* bits, const float* caps, const float* pi, const int16_t* perm, const int32_t* idx,
                     const int32_t* sym, int32_t count, int32_t board_size, int32_t history_len, int32_t row_bytes,
                     float* out_states, float* out_pi, void* stream);

/* The inverse direction: to_tensor() planes (float32 [count][2H+5][n][n], e.g. from az_rules_features) into the
 * store's compact form, `bits` [count][row_bytes] and `caps` [count][2] (go.py:265-304 planes; numpy.packbits order). */
int az_pack_planes(const float* planes, int32_t count, int32_t board_size, int32_t history_len, int32_t row_bytes, uint8_t* bits,
                   float* caps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AZAMD_H */
