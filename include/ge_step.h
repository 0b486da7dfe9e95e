/* ge_step.h — C ABI of the MI355X batch room-phase stepper (libge_step.so).
 *
 * Drop-in boundary for ONE path of liruihan000/game_engine: the per-turn loop
 *   InitialRouterNode -> BotBehaviorNode -> PhaseNode -> RefereeNode -> ActionExecutor
 *   (reference agent/game_agent_v2.py:198/468/987/619/1243, graph :1571-1587;
 *    newer fused form agent/game_agent_v3.py:205/448/540)
 * which the reference runs one room per LangGraph thread (src/app/api/copilotkit/route.ts:22-47)
 * with an LLM call per node.  Here the same turn is applied to a BATCH of independent rooms
 * on the GPU with the LLM replaced by the fixed policy of POLICY.md.
 *
 * Conventions:
 *   - plain C, no torch / STL types; every function returns 0 or a negative ge_status;
 *     nothing throws or aborts across this boundary; HIP errors map to GE_ERR_HIP.
 *   - the caller owns every host buffer it passes; the library owns device memory behind
 *     the opaque ge_batch handle.
 *   - a handle is not thread-safe: one handle per host thread / GPU.  Every call leaves the
 *     calling thread's current HIP device as it found it.
 *   - ge_batch_step is asynchronous on the given stream; read / summary / sync synchronise.
 *     Consecutive ge_batch_step calls may name different streams: the library orders each call
 *     behind the previous one with an event, and every synchronising call waits for the stream
 *     of the most recent step (which, by that ordering, is behind all earlier ones).
 *   - there is NO CPU fallback: without a HIP device ge_batch_create fails with GE_ERR_NO_DEVICE.
 */
#ifndef GE_STEP_H
#define GE_STEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (a host checks ge_abi_version() == GE_ABI_VERSION; a ge_game_table carries the version it was compiled by):
 *   2  stream ordering, device-side reset, batched injection     3  ge_game_table.field_names (schema binding)
 *   4  device group (ge_group_*, GE_ERR_COMM); ge_batch_write_rooms became all-or-nothing and REFUSES (GE_ERR_ARG) a view whose
 *      `pack` / player count is not the segment's or whose phase ids name no row of its table (until then unknown ids were
 *      silently stored as row 0) - a caller that zero-initialises views must set `pack`
 *   5  ge_group_partition + ge_batch_create_shard (the group's sharding arithmetic for hosts that place shards themselves); ge_last_rejected_room;
 *      mixed and generic batches get single-turn kernel builds; the Werewolf x 12 deal side plane is allocated on first use */
#define GE_ABI_VERSION 5
#define GE_MAX_PHASES 32
#define GE_MAX_PLAYERS 12
#define GE_MAX_SEGMENTS 4
#define GE_MAX_TERMS 4
#define GE_MAX_CLAUSES 4
#define GE_MAX_BRANCHES 4
#define GE_NAME_LEN 64
#define GE_MAX_SLOTS 12

typedef enum ge_status {
    GE_OK = 0,
    GE_ERR_ARG = -1,        /* null pointer, bad size, out-of-range value */
    GE_ERR_DSL = -2,        /* the DSL cannot be compiled (message in the err buffer) */
    GE_ERR_NO_DEVICE = -3,  /* no usable HIP device: the product has no CPU path */
    GE_ERR_HIP = -4,        /* a HIP runtime call failed (ge_last_hip_error) */
    GE_ERR_NOMEM = -5,
    GE_ERR_RANGE = -6,      /* room range outside the batch / turn counter would overflow */
    GE_ERR_UNSUPPORTED = -7,/* valid DSL feature the kernels do not implement yet; no RCCL to load for a device group */
    GE_ERR_COMM = -9        /* an RCCL call failed (ge_last_comm_error); -8 is the N-API host's GE_BUSY */
} ge_status;

/* rule packs: which declared player_states schema the game uses
 * (reference games/werewolf-(mafia).yaml:21-72, games/two-truths-and-a-lie.yaml declaration) */
enum { GE_PACK_WEREWOLF = 1, GE_PACK_TWO_TRUTHS = 2 };
/* State slots of the packs, in ge_room_view.players[] column order (0..8), then the slots that are not columns.
 * A generated DSL names its fields itself: each slot binds to the name the declaration uses for it
 * (ge_game_table.field_names; e.g. the reference's earlier Werewolf draft, game_draft/werewolf-(mafia).yaml:30-75,
 * says has_night_action / known_alignments / wolf_chat_enabled and declares no selected_target_id).  A slot the
 * DSL does not declare still exists in the record; it is just not part of the room's player_states. */
enum { GE_WW_ROLE = 0, GE_WW_TEAM, GE_WW_IS_ALIVE, GE_WW_ROLE_REVEALED, GE_WW_CAN_VOTE, GE_WW_HAS_SECRET_ROLE,
       GE_WW_NIGHT_ELIGIBLE,      /* night_action_eligible | has_night_action */
       GE_WW_NIGHT_SUBMITTED, GE_WW_SELECTED_TARGET,
       GE_WW_DET_MEMORY,          /* investigated_alignments | known_alignments (ge_room_view.det) */
       GE_WW_WOLF_CHAT,           /* wolf_chat_enabled: derived, = (team == werewolves) (POLICY.md 3a) */
       GE_WW_SLOTS };
enum { GE_TT_IS_SPEAKER = 0, GE_TT_STATEMENTS_SUBMITTED, GE_TT_LIE_INDEX, GE_TT_LIE_REVEALED, GE_TT_CAN_VOTE,
       GE_TT_VOTE_CHOICE, GE_TT_HAS_VOTED, GE_TT_TOTAL_SCORE, GE_TT_ROUNDS_AS_SPEAKER,
       GE_TT_STATEMENTS,          /* text the record does not carry; follows statements_submitted */
       GE_TT_SLOTS };
/* completion_criteria.type (dsl_phases_generation_prompt.txt:106-150) */
enum { GE_COMP_UI = 0, GE_COMP_TIMER = 1, GE_COMP_ACTION = 2 };
/* what a bot action in a player_action phase means (bot_behavior_system_prompt.txt:21-56) */
enum { GE_ACT_NONE = 0, GE_ACT_WOLF_TARGET, GE_ACT_DOCTOR_PROTECT, GE_ACT_DETECTIVE, GE_ACT_DAY_VOTE,
       GE_ACT_TT_STATEMENTS, GE_ACT_TT_LIE, GE_ACT_TT_VOTE };
/* what the Referee applies when a phase is entered (referee_system_prompt_2.txt:1-8,19-22,75-82) */
enum { GE_EFF_NONE = 0, GE_EFF_ASSIGN_ROLES, GE_EFF_NIGHT_BEGIN, GE_EFF_NIGHT_RESOLVE, GE_EFF_DAY_RESOLVE,
       GE_EFF_TT_ROUND_START, GE_EFF_TT_REVEAL, GE_EFF_TT_SCORE };
/* resolver bound to a natural-language next_phase key (ww:435-447, tt "Check Round Progress") */
enum { GE_RES_ALWAYS = 0, GE_RES_WOLVES_ZERO, GE_RES_WOLVES_GE_VILLAGERS, GE_RES_FOLLOWS_DAY,
       GE_RES_FOLLOWS_NIGHT, GE_RES_ALL_ROUNDS_DONE, GE_RES_OTHERWISE };

/* numeric player fields a target condition may compare (declared `num` fields of the rule packs:
 * werewolf selected_target_id; two-truths lie_index, vote_choice, total_score, rounds_as_speaker) */
enum { GE_NUM_SELECTED_TARGET = 0, GE_NUM_LIE_INDEX = 1, GE_NUM_VOTE_CHOICE = 2, GE_NUM_TOTAL_SCORE = 3,
       GE_NUM_ROUNDS_AS_SPEAKER = 4 };
enum { GE_LIT_NONE = 0, GE_LIT_BASE = 1, GE_LIT_NUM = 2 };

/* One literal of a target condition in clause form (OR of AND-clauses).  The grammar is the one the
 * reference's DSL generator is told to write (agent/prompt/dsl_phases_generation_prompt.txt:120-132):
 *   player.<field> ==|!=|<|<=|>|>= <value>,  player.<field> in [..] / not in [..],  joined by and / or.
 * GE_LIT_BASE: the player has ANY of the base predicates in the bit set `bases` (== / != / in over booleans
 * and enums, POLICY.md §3 numbering); GE_LIT_NUM: lo <= player.<num_field> <= hi (lo > hi: never true).
 * `neg` inverts the literal. */
typedef struct ge_literal {
    uint8_t kind;                         /* GE_LIT_* */
    uint8_t neg;
    uint8_t num_field;                    /* GE_NUM_* (GE_LIT_NUM) */
    uint8_t pad;
    uint16_t bases;                       /* GE_LIT_BASE: bit b = base predicate b */
    uint8_t lo, hi;                       /* GE_LIT_NUM */
} ge_literal;

/* One DSL phase, compiled.  Replaces what the LLM reads out of dsl['phases'][id] each turn
 * (v2:1057, 1087-1103). */
typedef struct ge_phase_row {
    int32_t phase_id;                     /* DSL id (0..16, 99, ...) */
    uint8_t completion;                   /* GE_COMP_* */
    uint8_t act;                          /* GE_ACT_*  (GE_COMP_ACTION phases) */
    uint8_t effect;                       /* GE_EFF_*  applied when this phase is entered */
    uint8_t n_terms;                      /* target_players.condition: AND of terms */
    uint8_t term_base[GE_MAX_TERMS];      /* base predicate index inside the pack (POLICY.md) */
    uint8_t term_neg[GE_MAX_TERMS];       /* 1: the term is "== false" / "!=" */
    uint8_t n_branches;                   /* 0: terminal phase (next_phase: null) */
    uint8_t br_res[GE_MAX_BRANCHES];      /* GE_RES_*, evaluated in DSL order, first match wins */
    uint8_t br_target[GE_MAX_BRANCHES];   /* dense row index of the successor */
    uint8_t pad[3];
    char name[GE_NAME_LEN];               /* phases.<id>.name, UTF-8, truncated */
    /* target_players.condition in clause form: always filled.  `generic` = 0: the condition is the plain
     * conjunction term_base / term_neg above (what both shipped games use; the kernels' fast path);
     * 1: only the clause form describes it (or / in [..] with several values / numeric comparisons). */
    uint8_t generic;
    uint8_t n_clauses;                    /* 0: no condition (every living player) */
    uint8_t clause_len[GE_MAX_CLAUSES];
    uint8_t pad2[2];
    ge_literal clause[GE_MAX_CLAUSES][GE_MAX_TERMS];
} ge_phase_row;

/* A compiled game (one YAML file).  Host-visible so callers may inspect or build one by hand. */
typedef struct ge_game_table {
    int32_t abi_version;
    int32_t pack;                         /* GE_PACK_* */
    int32_t n_phases;
    int32_t rounds;                       /* two-truths: agreed speaking turns per player */
    int32_t min_players;                  /* declaration.min_players */
    uint8_t init_fields[12];              /* player_states_template, canonical field order */
    char role_names[5][GE_NAME_LEN];      /* werewolf: "", Villager, Werewolf, Doctor, Detective as declared */
    char field_names[GE_MAX_SLOTS][GE_NAME_LEN];  /* slot (GE_WW_* / GE_TT_*) -> declared field name, "" = not declared */
    ge_phase_row rows[GE_MAX_PHASES];
} ge_game_table;

/* Compiles a game DSL given as JSON text (the host parses YAML with its own loader, as the
 * reference does: yaml.safe_load in agent/tools/utils.py:572, js-yaml in
 * src/app/api/games/initialize-players/route.ts).  Replaces the LLM's reading of the DSL.
 * `err` (may be NULL) receives a NUL-terminated message on GE_ERR_DSL. */
int ge_table_compile_json(const char *dsl_json, size_t len, int rounds, ge_game_table *out,
                          char *err, size_t err_cap);

/* Canonical, layout-independent view of one room: the integer projection of the reference's
 * AgentState (v2:97-117): current_phase_id, phase history tail, player_states fields.
 * players[i] = player id i+1; field order per pack:
 *   werewolf : role team is_alive role_revealed can_vote has_secret_role night_action_eligible
 *              night_action_submitted selected_target_id acted choice
 *   two-truths: is_speaker statements_submitted lie_index lie_revealed can_vote vote_choice
 *              has_voted total_score rounds_as_speaker acted choice
 * (acted/choice = this visit's latest logged action per player, i.e. the playerActions log
 *  of bt:285-344 reduced to what later turns read.)
 * det[i]: the Detective's investigated_alignments for player i+1: 0 unknown, 1 villagers, 2 werewolves. */
typedef struct ge_room_view {
    int32_t phase_id;
    int32_t prev_phase_id;
    int32_t end_turn;                     /* turn in which a terminal phase was entered (saturates at 65534), else -1 */
    int32_t games;                        /* GE_FLAG_RESTART: games this slot completed before the current one */
    uint8_t phase0_done;                  /* phase-0 guard of v2:1025-1052 already taken */
    uint8_t n_players;
    uint8_t pack;
    uint8_t pad;
    uint8_t players[16][12];
    uint8_t det[16];
} ge_room_view;

/* What one turn of one room logged and decided (GE_FLAG_TRACE). */
typedef struct ge_turn_event {
    uint32_t turn;
    int32_t from_phase_id;                /* current_phase_id when the turn started */
    int32_t to_phase_id;                  /* ... when it ended (== from: no transition) */
    uint16_t acted_now;                   /* bit i: player i+1 logged an action this turn (BotBehaviorNode) */
    uint8_t restarted;                    /* GE_FLAG_RESTART: the slot was re-initialised at the start of this turn */
    uint8_t pad;
    uint8_t choice[16];                   /* the choice each of those players logged (player id / statement no.) */
} ge_turn_event;

typedef struct ge_segment_desc {
    const ge_game_table *table;           /* copied at create; need not outlive the call */
    uint32_t n_players;                   /* werewolf 4..12, two-truths 3..12 */
    uint32_t human_mask;                  /* bit i: player i+1 is driven by the host (a human), not by the bot
                                             policy: BotBehaviorNode never acts for it (the reference excludes
                                             player 1, bot_behavior_system_prompt.txt:3) and a phase waits for its
                                             action (ge_batch_inject_action).  0 = all bots. */
    uint64_t n_rooms;
} ge_segment_desc;

/* ge_batch_desc.flags */
#define GE_FLAG_NONE 0u
/* steady state: a room that starts a turn in a terminal phase is first re-initialised to the
 * DSL template (a new game on the same slot; the turn counter and hence the RNG stream keep
 * running).  Off: terminal phases are absorbing, as in the reference. */
#define GE_FLAG_RESTART 1u
/* keep, for every room, one ge_turn_event per turn of the most recent ge_batch_step call (which
 * must then not exceed max_fuse turns): what the turn logged and decided, so that a host can
 * surface it as the reference's backend tool calls (update_player_actions / set_next_phase /
 * update_player_state, agent/tools/backend_tools.py:10-157).  Costs 16 B of HBM writes per room-turn. */
#define GE_FLAG_TRACE 2u

typedef struct ge_batch_desc {
    uint64_t seed;
    uint64_t first_room;                  /* global index of this batch's room 0 (sharding: the RNG is
                                             keyed by GLOBAL room index, so results do not depend on
                                             how rooms are split over GPUs) */
    uint32_t n_segments;                  /* rooms are grouped by game: one segment per (table, N) */
    uint32_t flags;
    int32_t device;                       /* HIP device ordinal */
    uint32_t max_fuse;                    /* turns fused into one launch (state stays in registers);
                                             0 = library default, 1 = one launch per turn */
    ge_segment_desc seg[GE_MAX_SEGMENTS];
} ge_batch_desc;

typedef struct ge_batch ge_batch;

/* Aggregate over the rooms of a batch (what the cross-shard all-gather exchanges).
 * Every field is a sum over rooms, so shard summaries add up to the whole-job summary. */
typedef struct ge_summary {
    uint64_t rooms;
    uint64_t finished;                    /* rooms in a terminal phase */
    uint64_t village_wins, wolf_wins;     /* werewolf rooms finished with no / some wolves alive */
    uint64_t alive_players;
    uint64_t sum_end_turn;                /* over finished rooms */
    uint64_t end_turn_hist[16];           /* finished rooms by end_turn / 8 (last bucket open) */
    uint64_t score_hist[16];              /* two-truths: players by total_score (last bucket open) */
    uint64_t checksum;                    /* sum over rooms of hash(global room index, packed state) */
    uint64_t turn;                        /* turns stepped so far */
    uint64_t games_recycled;              /* GE_FLAG_RESTART: finished games whose slot was re-initialised */
} ge_summary;

/* Creates the batch with every room in the DSL's initial state (player_states_template,
 * phase 0): the batched InitialRouterNode init of v2:255-289 + utils.py:584-653. */
int ge_batch_create(const ge_batch_desc *desc, ge_batch **out);

/* Advances EVERY room by n_turns turns (one turn = one graph run of the reference, §3.1).
 * `hip_stream` is a hipStream_t (NULL = the default stream).  Asynchronous. */
int ge_batch_step(ge_batch *b, uint32_t n_turns, void *hip_stream);

/* Back to the initial state and turn 0 (same seed, same rooms): a device-side fill of every room
 * record with the DSL's template state.  Synchronises. */
int ge_batch_reset(ge_batch *b);

/* Sets the turn counter (0 .. 2^32-1).  The RNG is keyed by (global room, turn) and end_turn and
 * the event trace are turn-based, so a checkpoint is (room records, turn): restore = create the
 * batch, ge_batch_write_rooms (or a copy into ge_batch_state), ge_batch_set_turn.  What the
 * reference's LangGraph checkpointer keeps per thread besides the state (the run counter).  Synchronises. */
int ge_batch_set_turn(ge_batch *b, uint64_t turn);     /* (also drops the records' prepared-deal caches: a raw restore may come from another seed) */

int ge_batch_sync(ge_batch *b);
int ge_batch_turn(const ge_batch *b, uint64_t *turn);
int ge_batch_n_rooms(const ge_batch *b, uint64_t *n_rooms);

/* Copies `count` rooms starting at local index `first` into dst (cap_bytes >= count*sizeof(ge_room_view)).
 * Synchronises.  Room order: segment 0's rooms, then segment 1's, ...
 * Only the packed records (32 - 48 B per room) cross PCIe, through a pinned staging buffer the batch keeps; the views are built
 * from them by the host cores - from 16 384 rooms on by several threads (at most 16, fewer if the process's CPU affinity or the
 * environment variable GE_IO_THREADS says so).  1 048 576 Werewolf x 8 rooms: 5.7 ms (profiles/r03_host_io.txt). */
int ge_batch_read_rooms(ge_batch *b, uint64_t first, uint64_t count, ge_room_view *dst, size_t cap_bytes);

/* Overwrites rooms from canonical views (checkpoint restore, tests of hand-built states).  All views are checked first
 * (n_players and pack of the segment the room lies in, phase_id / prev_phase_id naming phases of that segment's table - the
 * reference never stores an id outside dsl['phases'] either, v2:1173-1191 - werewolf role classes 0 .. 4): GE_ERR_ARG leaves
 * every room as it was.  Threads as above. */
int ge_batch_write_rooms(ge_batch *b, uint64_t first, uint64_t count, const ge_room_view *src);

/* Logs an action of a host-driven player between turns, exactly as if the player had acted in the
 * room's current phase: the action joins this visit's log (acted / choice) and the Referee's record
 * effect is applied (POLICY.md §3 "record").  What the reference does with a human's message at the
 * start of the next graph run (agent/tools/utils.py:310-358 -> bt:285-344).  `choice` is a player id
 * (werewolf) or a statement number (two-truths); GE_ERR_ARG if the player is not a target of the
 * current phase, has already acted, or the choice is out of range.  Synchronises. */
int ge_batch_inject_action(ge_batch *b, uint64_t room, uint32_t player_id, uint32_t choice);

/* The same for n actions at once (a batch of rooms with human seats): one kernel, one thread per
 * distinct room, which applies that room's actions in input order.  rooms[k] / player_ids[k] /
 * choices[k] describe action k; status[k] (status may be NULL) receives GE_OK or the reason it was
 * refused (GE_ERR_ARG as for ge_batch_inject_action, GE_ERR_RANGE for a room outside the batch).  A refused
 * action changes nothing; the others are applied.  Returns GE_OK if every action was applied, else the
 * status of the first refused one.  Synchronises. */
int ge_batch_inject_actions(ge_batch *b, uint64_t n, const uint64_t *rooms, const uint32_t *player_ids,
                            const uint32_t *choices, int32_t *status);

/* GE_FLAG_TRACE: events of the most recent ge_batch_step call, dst[(room - first) * *n_turns + t].
 * cap_bytes >= count * n_turns * sizeof(ge_turn_event).  Synchronises. */
int ge_batch_read_events(ge_batch *b, uint64_t first, uint64_t count, uint32_t *n_turns,
                         ge_turn_event *dst, size_t cap_bytes);

/* Device-side reduction of the whole batch.  Synchronises. */
int ge_batch_summary(ge_batch *b, ge_summary *out);

/* Raw packed state of one segment in HBM (for checkpoint = plain D2H copy, or zero-copy wrapping
 * by the host framework).  bytes_per_room is the algorithmic record size. */
int ge_batch_state(ge_batch *b, uint32_t segment, void **dev_ptr, size_t *bytes, uint32_t *bytes_per_room);

/* Timing of the kernels launched by the most recent ge_batch_step calls since the last reset,
 * measured with hipEvents on the stream they were launched on. */
int ge_batch_set_timing(ge_batch *b, int on);     /* off by default: no events are recorded */
int ge_batch_kernel_time(ge_batch *b, int reset, double *total_ms, uint64_t *launches);

void ge_batch_destroy(ge_batch *b);

/* ---- device group: ONE host process, N GPUs of a node (SURVEY.md 8(e) process model; what a Node addon needs).
 * Rooms never interact - the reference runs one LangGraph thread per room (src/app/api/copilotkit/route.ts:24-37) and has
 * no collective at all (agent/requirements.txt:1-11) - so the group shards rooms and steps the devices concurrently with no
 * exchange; the single collective of the whole job is ONE ncclAllGather (RCCL over xGMI) of the per-device ge_summary.
 *
 * `desc` describes the WHOLE job (desc->device is ignored): device i of n gets the i-th of n contiguous parts of every
 * segment, and every room keeps the global index it has in a single batch of the same desc - results are identical to
 * that batch's, whatever n is.  `devices` are distinct HIP ordinals (duplicates: GE_ERR_ARG); every segment needs at
 * least n rooms.  The group owns one stream per device.  RCCL is loaded at run time (librccl.so.1) when the first group
 * is created - GE_ERR_UNSUPPORTED if there is none; hosts that never create a group never load it.
 * Not thread-safe, like a ge_batch.  The multi-PROCESS form (one rank per GPU, torch.distributed) is game_engine_amd/dist.py. */
typedef struct ge_group ge_group;
/* The group's sharding, for a host that places the shards itself (several per device, or devices of its own choosing without RCCL)
 * - and what ge_group_create does internally: part `part` of `n_parts` takes the part-th of n_parts contiguous parts of every
 * segment of `desc` (the whole job; every segment needs >= n_parts rooms).  *shard = desc with those room counts;
 * seg_first[0 .. GE_MAX_SEGMENTS) = the global index of the part's first room of each segment (what the RNG is keyed by).
 * Pure host arithmetic: no device is touched.  ge_batch_create_shard(shard, seg_first, &b) then creates that part as an ordinary
 * batch (shard->device says where); n such batches hold, room for room, what ONE batch of `desc` holds, and their summaries add up
 * to its summary (checksums and histograms are sums over rooms). */
int ge_group_partition(const ge_batch_desc *desc, int n_parts, int part, ge_batch_desc *shard, uint64_t *seg_first);
int ge_batch_create_shard(const ge_batch_desc *shard, const uint64_t *seg_first, ge_batch **out);
int ge_group_create(const ge_batch_desc *desc, const int *devices, int n_devices, ge_group **out);
int ge_group_size(const ge_group *g);                           /* number of devices, or GE_ERR_ARG */
int ge_group_shard(ge_group *g, int i, ge_batch **out);         /* borrow device i's batch (read_rooms, inject, events ...); owned by the group */
int ge_group_step(ge_group *g, uint32_t n_turns);               /* every shard, asynchronous, each on its device's stream */
int ge_group_sync(ge_group *g);
/* Per-device reductions, one ncclAllGather of the n ge_summary records on the devices' streams, then the sum (every
 * field is a sum over rooms; `turn` is common).  Synchronises. */
int ge_group_summary(ge_group *g, ge_summary *out);
void ge_group_destroy(ge_group *g);
int ge_last_comm_error(void);             /* ncclResult_t of the last GE_ERR_COMM on this thread */

const char *ge_strerror(int status);
int ge_last_hip_error(void);              /* hipError_t of the last GE_ERR_HIP on this thread */
uint64_t ge_last_rejected_room(void);     /* ge_batch_write_rooms returned GE_ERR_ARG for a view that does not fit its segment: the index (in the batch) of
                                             the first such room, on this thread; ~0 if none yet (ABI 5) */
int ge_abi_version(void);
int ge_device_count(void);                /* number of HIP devices, 0 if none; never fails */

#ifdef __cplusplus
}
#endif
#endif /* GE_STEP_H */
