// Types for room_service.js — the single-room (one LangGraph thread) drop-in.
import { RoomState, ToolCall, FrontendToolCall } from './index';

export interface RoomPlayer { name?: string; gamePlayerId?: number; /** false marks a human seat (host-driven) */ isBot?: boolean; }
/** AgentState as the frontend syncs it (src/lib/canvas/types.ts:338-360), log-shaped parts included. */
export interface AgentStateView {
  gameName: string; current_phase_id: number; current_phase_name: string;
  player_states: Record<string, Record<string, unknown>>;
  playerActions: Record<string, { name: string; actions: Record<string, { action: string; timestamp: number; phase: string; id: string }> }>;
  phase_history: { phase_id: number; phase_name: string }[];
  game_notes: string[];
}
export interface TurnResult { state: AgentStateView; toolCalls: ToolCall[]; uiCalls: FrontendToolCall[]; }
export class RoomService {
  constructor(opts?: { gamesDir?: string; seed?: bigint | number; device?: number });
  createRoom(opts: { threadId: string; gameName: string; players: RoomPlayer[]; dsl?: object; /** global room index the RNG is keyed by (default: hash of the thread id) */ roomIndex?: number | bigint }): AgentStateView;
  /** Requests of one thread are served strictly one after the other. */
  humanAction(threadId: string, playerId: number, choice: number): Promise<AgentStateView>;
  /** One message of the browser, as the reference's graph reads it (page.tsx:272-275, 302-305, 341-349, 2774, 2843, 2962;
   *  POLICY.md 3b): chat plays no turn; anything else plays one, after a game message was logged under Player 1 and - where it
   *  is a valid action of a host-driven seat - applied. */
  handleMessage(threadId: string, text: string, items?: { id: string; type: string }[]): Promise<TurnResult & { played: boolean; kind: 'chat' | 'control' | 'action' }>;
  /** items: the frontend's canvas items (AgentState.items), for clearCanvas's exemptList */
  continueRoom(threadId: string, items?: { id: string; type: string }[]): Promise<TurnResult>;
  /** Forget a thread and free its device memory; resolves false for an unknown thread. */
  close(threadId: string): Promise<boolean>;
  serve(port?: number): Promise<import('http').Server>;
}
export function roomIndexOf(threadId: string): bigint;
export type { RoomState };
