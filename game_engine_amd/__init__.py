"""game_engine_amd — MI355X-native batch room-phase stepper.

Host-side Python mirror of the one reference path this repository accelerates: the per-turn
loop of liruihan000/game_engine (agent/game_agent_v2.py:1571-1587).  All compute is in
libge_step.so (HIP, gfx950) behind the C ABI of include/ge_step.h; there is no CPU fallback.
"""
from .stepper import (GameTable, RoomBatch, RoomGroup, GeError, load_dsl_by_gamename, initialize_player_states_from_dsl, library_path,
                      ROOM_VIEW_DTYPE, EVENT_DTYPE, WW_FIELDS, TT_FIELDS)

from .room_service import RoomService, room_index_of

__all__ = ["RoomService", "room_index_of", "GameTable", "RoomBatch", "RoomGroup", "GeError", "load_dsl_by_gamename", "initialize_player_states_from_dsl", "library_path",
           "ROOM_VIEW_DTYPE", "EVENT_DTYPE", "WW_FIELDS", "TT_FIELDS"]
