// Type declarations for the Node/TypeScript host of the MI355X batch stepper.
// The state shapes mirror the reference's TS AgentState (src/lib/canvas/types.ts:338-360).

export interface WerewolfPlayerState {
  role: string; team: '' | 'villagers' | 'werewolves'; is_alive: boolean; role_revealed: boolean;
  can_vote: boolean; has_secret_role: boolean; night_action_eligible: boolean;
  night_action_submitted: boolean; selected_target_id: number;
  investigated_alignments: Record<string, string>;
}
export interface TwoTruthsPlayerState {
  is_speaker: boolean; statements_submitted: boolean; lie_index: number; lie_revealed: boolean;
  can_vote: boolean; vote_choice: number; has_voted: boolean; total_score: number; rounds_as_speaker: number;
}
export interface RoomState {
  current_phase_id: number;
  current_phase_name: string;
  previous_phase_id: number;
  end_turn: number;                 // -1 while the game runs
  games: number;
  /** exactly the fields the DSL declares, under the DSL's own names (GameTable.info.fieldNames) */
  player_states: Record<string, WerewolfPlayerState | TwoTruthsPlayerState | Record<string, unknown>>;
  pack: number;                     // 1 werewolf, 2 two-truths
  slots: unknown[][];               // per player: one value per slot of the pack (GE_WW_* / GE_TT_* order), declared or not
  acted: number[];                  // this visit's action log, per player
  choice: number[];
}
export interface Summary {
  rooms: bigint; finished: bigint; village_wins: bigint; wolf_wins: bigint; alive_players: bigint;
  sum_end_turn: bigint; end_turn_hist: bigint[]; score_hist: bigint[]; checksum: bigint; turn: bigint;
  games_recycled: bigint;
}
export interface TurnEvent {
  turn: number; from_phase_id: number; to_phase_id: number; acted_now: number; restarted: number; choice: number[];
}
export interface ToolCall { name: 'update_player_actions' | 'set_next_phase' | 'update_player_state' | 'add_game_note'; args: Record<string, unknown>; }
export interface PhaseInfo { id: number; name: string; completion: number; act: number; effect: number; }

export class GameTable {
  constructor(dsl: object, rounds?: number);
  static fromGamename(gamename: string, gamesDir?: string, rounds?: number): GameTable;
  /** fieldNames: slot (include/ge_step.h GE_WW_* / GE_TT_*) -> the DSL's own field name, "" = not declared */
  readonly info: { pack: number; rounds: number; minPlayers: number; roleNames: string[]; fieldNames: string[]; phases: PhaseInfo[] };
  phaseName(id: number): string;
}
export interface Segment { table: GameTable; nPlayers: number; nRooms: number; /** bit i: player i+1 is driven by the host (a human) */ humanMask?: number; }
export interface BatchOptions {
  segments: Segment[]; seed?: bigint | number; firstRoom?: bigint | number; device?: number;
  maxFuse?: number; restart?: boolean; trace?: boolean;
}
export class RoomBatch {
  constructor(opts: BatchOptions);
  readonly nRooms: number;
  /** Async steps of one batch are chained; a synchronous call made while one is in flight throws GE_BUSY. */
  step(nTurns?: number): Promise<number>;
  whenIdle<T>(fn: () => T): Promise<T>;
  stepSync(nTurns?: number): number;
  reset(): void;
  /** Checkpoint = readRoomsRaw + the turn; restore = writeRoomsRaw + setTurn into a fresh batch. */
  readRoomsRaw(first: number, count: number): ArrayBuffer;
  writeRoomsRaw(first: number, buffer: ArrayBuffer): void;
  setTurn(turn: number | bigint): void;
  close(): void;
  injectAction(room: number, playerId: number, choice: number): void;
  /** One kernel for many host-driven players' actions; per-action status, 0 = applied. */
  injectActions(rooms: ArrayLike<number | bigint>, playerIds: ArrayLike<number>, choices: ArrayLike<number>): Int32Array;
  readRoom(room: number): RoomState;
  readRooms(first: number, count: number): RoomState[];
  readEvents(first: number, count: number): TurnEvent[][];
  summary(): Summary;
}
/** One Node process, several GPUs: device d owns the global rooms [firstRoom + d*R, firstRoom + (d+1)*R). */
export class ShardedBatch {
  constructor(opts: { segments: Segment[]; devices: number[]; seed?: bigint; firstRoom?: bigint; maxFuse?: number; restart?: boolean; trace?: boolean });
  readonly nRooms: number;
  readonly roomsPerDevice: number;
  readonly shards: RoomBatch[];
  step(nTurns?: number): Promise<bigint | number>;
  reset(): void;
  close(): void;
  readRoom(room: number): RoomState;
  injectAction(room: number, playerId: number, choice: number): void;
  summary(): Summary;
}
/** The native device group (ge_group_*): ONE Node process, N distinct GPUs.  `segments` = the WHOLE job, split per segment
 * over the devices with the global room indices of one RoomBatch; summary() = per-device reductions + ONE RCCL all-gather
 * inside libge_step.so.  (ShardedBatch is the host-side form: host-summed, also runs several shards on one device.) */
export class DeviceGroup {
  constructor(opts: { segments: Segment[]; devices: number[]; seed?: bigint; firstRoom?: bigint; maxFuse?: number; restart?: boolean; trace?: boolean });
  readonly nRooms: number;
  readonly devices: number[];
  step(nTurns?: number): Promise<void>;
  whenIdle<T>(fn: () => T): Promise<T>;
  summary(): Summary;
  /** room = index in segment-major order, as in one RoomBatch of the same segments */
  readRoom(room: number): RoomState;
  locate(room: number): [number, number, GameTable];   // via locateInShards
  close(): void;
}
export function turnToolCalls(table: GameTable, before: RoomState, after: RoomState, event: TurnEvent): ToolCall[];
/** What _execute_add_game_note appends: '<mark> <TYPE>: <content>' (backend_tools.py:175-198). */
export function formatNote(noteType: string, content: string): string;
/** playerActions / game_notes / phase_history / statements of one room, folded from each turn's tool calls as the reference's _execute_* would. */
export class RoomLog {
  constructor(table: GameTable, names: string[], gameName?: string);
  fold(calls: ToolCall[], after: RoomState, now?: number): void;
  agentState(room: RoomState): Record<string, unknown>;
}
export function loadDslByGamename(gamename: string, gamesDir?: string): object;
export function deviceCount(): number;
/** route.ts:62-70 file matching (case-insensitive, non-alphanumerics equal '-') */
export function findGameFile(gameName: string, gamesDir?: string): string | null;
export interface RoomPlayer { id?: string; name: string; isHost?: boolean; gamePlayerId?: string; }
/** POST /api/games/initialize-players (route.ts:83-166) without the HTTP layer */
export function initializePlayers(dsl: object, roomPlayers: RoomPlayer[]): { player_states: Record<string, Record<string, unknown>>; fallback_mode?: boolean; message?: string };
export function compileCriteria(expr: string): (player: Record<string, unknown>) => boolean;
export function audienceGroups(dsl: object, playerStates: Record<string, Record<string, unknown>>): Record<string, string[]>;
export interface FrontendToolCall { name: string; args: { audience_type?: boolean; audience_ids?: string[]; [k: string]: unknown }; }
/** Frontend tool calls of the room's current phase (ActionExecutor / UIUpdateNode without an LLM): every parameter the
 * handler requires (frontend_tools.json), nothing it does not declare. */
export function uiToolCalls(dsl: object, room: RoomState, opts?: { table?: GameTable; act?: number; turn?: number; deaths?: string[];
  items?: { id: string; type: string }[] }): FrontendToolCall[];
export function validateCall(call: FrontendToolCall): string[];
/** [part, index inside the part's batch, segment] of a room of the whole job after the group's sharding (ge_group_partition). */
export function locateInShards(segmentRooms: number[], nParts: number, room: number): [number, number, number];
