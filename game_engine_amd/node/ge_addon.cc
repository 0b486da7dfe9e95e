// ge_addon.cc — thin N-API binding of include/ge_step.h for the TypeScript/Node host.
//
// Where it sits in the reference: src/app/api/copilotkit/route.ts:22-47 builds a
// LangGraphAgent per request that forwards the room's AgentState over HTTP to the Python
// LangGraph server (agent/langgraph.json:5-8 -> game_agent_v2.py:graph).  A host that wants the
// GPU stepper instead calls this addon (see index.js / index.d.ts and INTEGRATION.md).
// No game logic here: every export is a 1:1 wrapper of a C-ABI entry point; stepping runs on the
// libuv pool (napi_create_async_work) so the event loop is never blocked.
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/ge_step.h"

namespace {

#define NAPI_OK(call)                                                        \
    do {                                                                     \
        if ((call) != napi_ok) {                                             \
            napi_throw_error(env, "GE_NAPI", "N-API call failed: " #call);   \
            return nullptr;                                                  \
        }                                                                    \
    } while (0)

napi_value throw_status(napi_env env, int st, const char *what, const char *detail = nullptr) {
    bool pending = false;
    if (napi_is_exception_pending(env, &pending) == napi_ok && pending) return nullptr;   // e.g. GE_BUSY from batch_arg
    std::string msg = std::string(what) + ": " + ge_strerror(st);
    if (detail && *detail) msg += std::string(" (") + detail + ")";
    char code[16];
    snprintf(code, sizeof code, "GE%d", st);
    napi_throw_error(env, code, msg.c_str());
    return nullptr;
}

bool get_u64(napi_env env, napi_value v, uint64_t *out) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return false;
    if (t == napi_bigint) {
        bool lossless;
        return napi_get_value_bigint_uint64(env, v, out, &lossless) == napi_ok;
    }
    if (t == napi_number) {
        double d;
        if (napi_get_value_double(env, v, &d) != napi_ok || d < 0) return false;
        *out = (uint64_t)d;
        return true;
    }
    return false;
}

bool get_prop_u64(napi_env env, napi_value obj, const char *name, uint64_t *out, uint64_t dflt) {
    bool has = false;
    napi_value v;
    *out = dflt;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return true;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return false;
    napi_valuetype t;
    napi_typeof(env, v, &t);
    if (t == napi_undefined || t == napi_null) return true;
    if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); *out = b; return true; }
    return get_u64(env, v, out);
}

// What a batch External holds.  The C handle is not thread-safe (ge_step.h), so while an async step owns
// it on a libuv worker (`busy`) every other entry point refuses it instead of racing, and the worker keeps
// a reference on the External so that the garbage collector cannot finalize it mid-step.
struct BatchBox {
    ge_batch *b = nullptr;
    bool busy = false;
};
void finalize_batch(napi_env, void *data, void *) {
    BatchBox *box = static_cast<BatchBox *>(data);
    if (box->b) ge_batch_destroy(box->b);
    delete box;
}
void finalize_table(napi_env, void *data, void *) { free(data); }

// compileTable(dslJson: string, rounds?: number): External<ge_game_table>
napi_value CompileTable(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 1) return throw_status(env, GE_ERR_ARG, "compileTable");
    size_t len = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], nullptr, 0, &len));
    std::string json(len, '\0');
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], &json[0], len + 1, &len));
    uint64_t rounds = 1;
    if (argc > 1 && !get_u64(env, argv[1], &rounds)) rounds = 1;
    ge_game_table *t = static_cast<ge_game_table *>(calloc(1, sizeof(ge_game_table)));
    char err[512];
    int st = ge_table_compile_json(json.data(), json.size(), (int)rounds, t, err, sizeof err);
    if (st != GE_OK) { free(t); return throw_status(env, st, "compileTable", err); }
    napi_value ext;
    NAPI_OK(napi_create_external(env, t, finalize_table, nullptr, &ext));
    return ext;
}

// tableInfo(table): { pack, rounds, minPlayers, roleNames[], fieldNames[], phases:[{id,name,completion,act,effect}] }
napi_value TableInfo(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_game_table *t = nullptr;
    if (argc < 1 || napi_get_value_external(env, argv[0], reinterpret_cast<void **>(&t)) != napi_ok || !t)
        return throw_status(env, GE_ERR_ARG, "tableInfo");
    napi_value out, phases, roles, v;
    NAPI_OK(napi_create_object(env, &out));
    NAPI_OK(napi_create_int32(env, t->pack, &v)); NAPI_OK(napi_set_named_property(env, out, "pack", v));
    NAPI_OK(napi_create_int32(env, t->rounds, &v)); NAPI_OK(napi_set_named_property(env, out, "rounds", v));
    NAPI_OK(napi_create_int32(env, t->min_players, &v)); NAPI_OK(napi_set_named_property(env, out, "minPlayers", v));
    NAPI_OK(napi_create_array_with_length(env, 5, &roles));
    for (uint32_t i = 0; i < 5; i++) {
        NAPI_OK(napi_create_string_utf8(env, t->role_names[i], NAPI_AUTO_LENGTH, &v));
        NAPI_OK(napi_set_element(env, roles, i, v));
    }
    NAPI_OK(napi_set_named_property(env, out, "roleNames", roles));
    napi_value fields;                     // slot (GE_WW_* / GE_TT_*) -> the DSL's own field name, "" = not declared
    NAPI_OK(napi_create_array_with_length(env, GE_MAX_SLOTS, &fields));
    for (uint32_t i = 0; i < GE_MAX_SLOTS; i++) {
        NAPI_OK(napi_create_string_utf8(env, t->field_names[i], NAPI_AUTO_LENGTH, &v));
        NAPI_OK(napi_set_element(env, fields, i, v));
    }
    NAPI_OK(napi_set_named_property(env, out, "fieldNames", fields));
    NAPI_OK(napi_create_array_with_length(env, t->n_phases, &phases));
    for (int i = 0; i < t->n_phases; i++) {
        napi_value ph;
        NAPI_OK(napi_create_object(env, &ph));
        NAPI_OK(napi_create_int32(env, t->rows[i].phase_id, &v)); NAPI_OK(napi_set_named_property(env, ph, "id", v));
        NAPI_OK(napi_create_string_utf8(env, t->rows[i].name, NAPI_AUTO_LENGTH, &v)); NAPI_OK(napi_set_named_property(env, ph, "name", v));
        NAPI_OK(napi_create_int32(env, t->rows[i].completion, &v)); NAPI_OK(napi_set_named_property(env, ph, "completion", v));
        NAPI_OK(napi_create_int32(env, t->rows[i].act, &v)); NAPI_OK(napi_set_named_property(env, ph, "act", v));
        NAPI_OK(napi_create_int32(env, t->rows[i].effect, &v)); NAPI_OK(napi_set_named_property(env, ph, "effect", v));
        NAPI_OK(napi_set_element(env, phases, i, ph));
    }
    NAPI_OK(napi_set_named_property(env, out, "phases", phases));
    return out;
}

// {seed, firstRoom, device, maxFuse, restart, trace, segments:[{table, nPlayers, nRooms, humanMask}]} -> ge_batch_desc; throws and returns false on a bad argument
bool parse_desc(napi_env env, napi_value obj, const char *what, ge_batch_desc *out) {
    ge_batch_desc &d = *out;
    memset(&d, 0, sizeof d);
    uint64_t dev = 0, fuse = 0, restart = 0, trace = 0;
    if (!get_prop_u64(env, obj, "seed", &d.seed, 0) || !get_prop_u64(env, obj, "firstRoom", &d.first_room, 0) ||
        !get_prop_u64(env, obj, "device", &dev, 0) || !get_prop_u64(env, obj, "maxFuse", &fuse, 0) ||
        !get_prop_u64(env, obj, "restart", &restart, 0) || !get_prop_u64(env, obj, "trace", &trace, 0)) {
        throw_status(env, GE_ERR_ARG, what);
        return false;
    }
    d.device = (int32_t)dev; d.max_fuse = (uint32_t)fuse;
    d.flags = (restart ? GE_FLAG_RESTART : GE_FLAG_NONE) | (trace ? GE_FLAG_TRACE : GE_FLAG_NONE);
    napi_value segs;
    uint32_t n = 0;
    bool is_arr = false;
    if (napi_get_named_property(env, obj, "segments", &segs) != napi_ok ||
        napi_is_array(env, segs, &is_arr) != napi_ok || !is_arr ||
        napi_get_array_length(env, segs, &n) != napi_ok || n == 0 || n > GE_MAX_SEGMENTS) {
        throw_status(env, GE_ERR_ARG, what, "segments");
        return false;
    }
    d.n_segments = n;
    for (uint32_t k = 0; k < n; k++) {
        napi_value sg, tv;
        uint64_t np = 0, nr = 0, hm = 0;
        ge_game_table *t = nullptr;
        if (napi_get_element(env, segs, k, &sg) != napi_ok ||
            napi_get_named_property(env, sg, "table", &tv) != napi_ok ||
            napi_get_value_external(env, tv, reinterpret_cast<void **>(&t)) != napi_ok || !t ||
            !get_prop_u64(env, sg, "nPlayers", &np, 0) || !get_prop_u64(env, sg, "nRooms", &nr, 0) ||
            !get_prop_u64(env, sg, "humanMask", &hm, 0)) {
            throw_status(env, GE_ERR_ARG, what, "segment");
            return false;
        }
        d.seg[k].table = t; d.seg[k].n_players = (uint32_t)np; d.seg[k].n_rooms = nr; d.seg[k].human_mask = (uint32_t)hm;
    }
    return true;
}

// createBatch({seed, firstRoom, device, maxFuse, restart, segments:[{table, nPlayers, nRooms}]}): External<ge_batch>
napi_value CreateBatch(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 1) return throw_status(env, GE_ERR_ARG, "createBatch");
    ge_batch_desc d;
    if (!parse_desc(env, argv[0], "createBatch", &d)) return nullptr;
    ge_batch *b = nullptr;
    int st = ge_batch_create(&d, &b);
    if (st != GE_OK) return throw_status(env, st, "createBatch");
    BatchBox *box = new BatchBox();
    box->b = b;
    napi_value ext;
    if (napi_create_external(env, box, finalize_batch, nullptr, &ext) != napi_ok) {
        ge_batch_destroy(b);
        delete box;
        napi_throw_error(env, "GE_NAPI", "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

BatchBox *box_arg(napi_env env, napi_value v) {
    BatchBox *box = nullptr;
    if (napi_get_value_external(env, v, reinterpret_cast<void **>(&box)) != napi_ok) return nullptr;
    return box;
}

// the handle of a batch that is alive and not owned by an async step; throws and returns null otherwise
ge_batch *batch_arg(napi_env env, napi_value v) {
    BatchBox *box = box_arg(env, v);
    if (!box || !box->b) return nullptr;
    if (box->busy) {
        napi_throw_error(env, "GE_BUSY", "an async step() of this batch is in flight: await it first (a ge_batch handle is not thread-safe)");
        return nullptr;
    }
    return box->b;
}

struct StepWork {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep;            // the External: alive until the worker is done
    BatchBox *box;
    uint32_t turns;
    int status;
};

void step_execute(napi_env, void *data) {
    StepWork *w = static_cast<StepWork *>(data);
    w->status = ge_batch_step(w->box->b, w->turns, nullptr);
    if (w->status == GE_OK) w->status = ge_batch_sync(w->box->b);
}

void step_complete(napi_env env, napi_status, void *data) {
    StepWork *w = static_cast<StepWork *>(data);
    w->box->busy = false;
    ge_batch *const batch_of_w = w->box->b;
    napi_value v;
    if (w->status == GE_OK) {
        uint64_t turn = 0;
        ge_batch_turn(batch_of_w, &turn);
        napi_create_double(env, (double)turn, &v);
        napi_resolve_deferred(env, w->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, ge_strerror(w->status), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &v);
        napi_reject_deferred(env, w->deferred, v);
    }
    napi_delete_async_work(env, w->work);
    napi_delete_reference(env, w->keep);          // last: from here on the External may be collected
    delete w;
}

// step(batch, nTurns): Promise<number /* turn counter */>
napi_value Step(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t turns = 1;
    if (!b || (argc > 1 && !get_u64(env, argv[1], &turns))) return throw_status(env, GE_ERR_ARG, "step");
    StepWork *w = new StepWork();
    w->box = box_arg(env, argv[0]); w->turns = (uint32_t)turns; w->status = GE_OK;
    napi_value promise, name;
    if (napi_create_reference(env, argv[0], 1, &w->keep) != napi_ok) { delete w; return throw_status(env, GE_ERR_ARG, "step"); }
    if (napi_create_promise(env, &w->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "ge_batch_step", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, nullptr, name, step_execute, step_complete, w, &w->work) != napi_ok) {
        napi_delete_reference(env, w->keep);
        delete w;
        return throw_status(env, GE_ERR_ARG, "step");
    }
    w->box->busy = true;
    if (napi_queue_async_work(env, w->work) != napi_ok) {
        w->box->busy = false;
        napi_delete_async_work(env, w->work);
        napi_delete_reference(env, w->keep);
        delete w;
        return throw_status(env, GE_ERR_ARG, "step");
    }
    return promise;
}

// stepSync(batch, nTurns): number
napi_value StepSync(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t turns = 1;
    if (!b || (argc > 1 && !get_u64(env, argv[1], &turns))) return throw_status(env, GE_ERR_ARG, "stepSync");
    int st = ge_batch_step(b, (uint32_t)turns, nullptr);
    if (st == GE_OK) st = ge_batch_sync(b);
    if (st != GE_OK) return throw_status(env, st, "stepSync");
    uint64_t turn = 0;
    ge_batch_turn(b, &turn);
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)turn, &v));
    return v;
}

// readRooms(batch, first, count): ArrayBuffer (count * sizeof(ge_room_view) bytes)
napi_value ReadRooms(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t first = 0, count = 0;
    if (!b || argc < 3 || !get_u64(env, argv[1], &first) || !get_u64(env, argv[2], &count))
        return throw_status(env, GE_ERR_ARG, "readRooms");
    void *data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)count * sizeof(ge_room_view), &data, &buf));
    int st = ge_batch_read_rooms(b, first, count, static_cast<ge_room_view *>(data), (size_t)count * sizeof(ge_room_view));
    if (st != GE_OK) return throw_status(env, st, "readRooms");
    return buf;
}

// injectAction(batch, room, playerId, choice): throws GE-1 if the action is not allowed
napi_value InjectAction(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t room = 0, player = 0, choice = 0;
    if (!b || argc < 4 || !get_u64(env, argv[1], &room) || !get_u64(env, argv[2], &player) || !get_u64(env, argv[3], &choice))
        return throw_status(env, GE_ERR_ARG, "injectAction");
    int st = ge_batch_inject_action(b, room, (uint32_t)player, (uint32_t)choice);
    if (st != GE_OK) return throw_status(env, st, "injectAction");
    return nullptr;
}

// injectActions(batch, rooms: BigUint64Array, playerIds: Uint32Array, choices: Uint32Array): Int32Array of per-action status
napi_value InjectActions(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    if (!b || argc < 4) return throw_status(env, GE_ERR_ARG, "injectActions");
    napi_typedarray_type tt[3];
    size_t len[3];
    void *data[3];
    for (int k = 0; k < 3; k++) {
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, argv[1 + k], &tt[k], &len[k], &data[k], &ab, &off) != napi_ok)
            return throw_status(env, GE_ERR_ARG, "injectActions", "typed arrays expected");
    }
    if (tt[0] != napi_biguint64_array || tt[1] != napi_uint32_array || tt[2] != napi_uint32_array || len[0] != len[1] || len[1] != len[2])
        return throw_status(env, GE_ERR_ARG, "injectActions", "BigUint64Array, Uint32Array, Uint32Array of equal length");
    void *st_data = nullptr;
    napi_value st_buf, st_arr;
    NAPI_OK(napi_create_arraybuffer(env, len[0] * sizeof(int32_t), &st_data, &st_buf));
    memset(st_data, 0, len[0] * sizeof(int32_t));
    const int st = ge_batch_inject_actions(b, len[0], static_cast<const uint64_t *>(data[0]), static_cast<const uint32_t *>(data[1]),
                                           static_cast<const uint32_t *>(data[2]), static_cast<int32_t *>(st_data));
    if (st != GE_OK) {
        // the return value is the first refused action's status - or a failure of the call itself (allocation, HIP error,
        // too many actions), which leaves the status array untouched: that one must not read as "all applied"
        bool any = false;
        for (size_t k = 0; k < len[0] && !any; k++) any = static_cast<const int32_t *>(st_data)[k] != 0;
        if (!any) return throw_status(env, st, "injectActions");
    }
    NAPI_OK(napi_create_typedarray(env, napi_int32_array, len[0], st_buf, 0, &st_arr));
    return st_arr;
}

// setTurn(batch, turn): restores a checkpoint's turn counter (the RNG is keyed by it)
napi_value SetTurn(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t turn = 0;
    if (!b || argc < 2 || !get_u64(env, argv[1], &turn)) return throw_status(env, GE_ERR_ARG, "setTurn");
    int st = ge_batch_set_turn(b, turn);
    if (st != GE_OK) return throw_status(env, st, "setTurn");
    return nullptr;
}

// writeRooms(batch, first, buffer: ArrayBuffer of ge_room_view): checkpoint restore
napi_value WriteRooms(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t first = 0;
    void *data = nullptr;
    size_t bytes = 0;
    if (!b || argc < 3 || !get_u64(env, argv[1], &first) || napi_get_arraybuffer_info(env, argv[2], &data, &bytes) != napi_ok ||
        bytes % sizeof(ge_room_view) != 0)
        return throw_status(env, GE_ERR_ARG, "writeRooms");
    int st = ge_batch_write_rooms(b, first, bytes / sizeof(ge_room_view), static_cast<const ge_room_view *>(data));
    if (st != GE_OK) return throw_status(env, st, "writeRooms");
    return nullptr;
}

// destroyBatch(batch): releases the device memory now instead of at garbage collection
napi_value DestroyBatch(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    BatchBox *box = argc >= 1 ? box_arg(env, argv[0]) : nullptr;
    if (!box) return throw_status(env, GE_ERR_ARG, "destroyBatch");
    if (box->busy) { napi_throw_error(env, "GE_BUSY", "destroyBatch while an async step() is in flight"); return nullptr; }
    if (box->b) ge_batch_destroy(box->b);
    box->b = nullptr;
    return nullptr;
}

// readEvents(batch, first, count): { nTurns, buffer: ArrayBuffer of count*nTurns ge_turn_event }
napi_value ReadEvents(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    uint64_t first = 0, count = 0;
    if (!b || argc < 3 || !get_u64(env, argv[1], &first) || !get_u64(env, argv[2], &count))
        return throw_status(env, GE_ERR_ARG, "readEvents");
    uint32_t nt = 0;
    int st = ge_batch_read_events(b, first, 0, &nt, nullptr, 0);
    if (st != GE_OK) return throw_status(env, st, "readEvents");
    void *data = nullptr;
    napi_value buf, out, v;
    const size_t bytes = (size_t)count * nt * sizeof(ge_turn_event);
    NAPI_OK(napi_create_arraybuffer(env, bytes, &data, &buf));
    st = ge_batch_read_events(b, first, count, &nt, static_cast<ge_turn_event *>(data), bytes);
    if (st != GE_OK) return throw_status(env, st, "readEvents");
    NAPI_OK(napi_create_object(env, &out));
    NAPI_OK(napi_create_uint32(env, nt, &v));
    NAPI_OK(napi_set_named_property(env, out, "nTurns", v));
    NAPI_OK(napi_set_named_property(env, out, "buffer", buf));
    return out;
}

// summary(batch): BigUint64Array-compatible ArrayBuffer of ge_summary words
napi_value Summary(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    if (!b) return throw_status(env, GE_ERR_ARG, "summary");
    void *data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_arraybuffer(env, sizeof(ge_summary), &data, &buf));
    int st = ge_batch_summary(b, static_cast<ge_summary *>(data));
    if (st != GE_OK) return throw_status(env, st, "summary");
    return buf;
}

napi_value Reset(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_batch *b = argc >= 1 ? batch_arg(env, argv[0]) : nullptr;
    if (!b) return throw_status(env, GE_ERR_ARG, "reset");
    int st = ge_batch_reset(b);
    if (st != GE_OK) return throw_status(env, st, "reset");
    return nullptr;
}

napi_value DeviceCount(napi_env env, napi_callback_info) {
    napi_value v;
    NAPI_OK(napi_create_int32(env, ge_device_count(), &v));
    return v;
}

napi_value RoomViewSize(napi_env env, napi_callback_info) {
    napi_value v;
    NAPI_OK(napi_create_int32(env, (int32_t)sizeof(ge_room_view), &v));
    return v;
}

// ---- device group: ONE Node process, N GPUs (ge_group_*, include/ge_step.h): rooms sharded over the devices, stepped
// concurrently, one RCCL all-gather of the per-device summaries inside the native library
struct GroupBox {
    ge_group *g = nullptr;
    bool busy = false;
};
void finalize_group(napi_env, void *data, void *) {
    GroupBox *box = static_cast<GroupBox *>(data);
    if (box->g) ge_group_destroy(box->g);
    delete box;
}
GroupBox *group_box(napi_env env, napi_value v) {
    GroupBox *box = nullptr;
    if (napi_get_value_external(env, v, reinterpret_cast<void **>(&box)) != napi_ok) return nullptr;
    return box;
}
ge_group *group_arg(napi_env env, napi_value v) {
    GroupBox *box = group_box(env, v);
    if (!box || !box->g) return nullptr;
    if (box->busy) {
        napi_throw_error(env, "GE_BUSY", "an async step() of this group is in flight: await it first (a ge_group handle is not thread-safe)");
        return nullptr;
    }
    return box->g;
}

// createGroup({seed, firstRoom, maxFuse, restart, trace, segments (the WHOLE job), devices: [0, 1, ...]}): External<ge_group>
napi_value CreateGroup(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 1) return throw_status(env, GE_ERR_ARG, "createGroup");
    ge_batch_desc d;
    if (!parse_desc(env, argv[0], "createGroup", &d)) return nullptr;
    napi_value devs;
    uint32_t n = 0;
    bool is_arr = false;
    if (napi_get_named_property(env, argv[0], "devices", &devs) != napi_ok || napi_is_array(env, devs, &is_arr) != napi_ok || !is_arr ||
        napi_get_array_length(env, devs, &n) != napi_ok || n == 0 || n > 64)
        return throw_status(env, GE_ERR_ARG, "createGroup", "devices");
    std::vector<int> devices(n);
    for (uint32_t i = 0; i < n; i++) {
        napi_value v;
        uint64_t x = 0;
        if (napi_get_element(env, devs, i, &v) != napi_ok || !get_u64(env, v, &x)) return throw_status(env, GE_ERR_ARG, "createGroup", "devices");
        devices[i] = (int)x;
    }
    ge_group *g = nullptr;
    int st = ge_group_create(&d, devices.data(), (int)n, &g);
    if (st != GE_OK) return throw_status(env, st, "createGroup");
    GroupBox *box = new GroupBox();
    box->g = g;
    napi_value ext;
    if (napi_create_external(env, box, finalize_group, nullptr, &ext) != napi_ok) {
        ge_group_destroy(g);
        delete box;
        napi_throw_error(env, "GE_NAPI", "napi_create_external failed");
        return nullptr;
    }
    return ext;
}

struct GroupStepWork {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep;
    GroupBox *box;
    uint32_t turns;
    int status;
};
void group_step_execute(napi_env, void *data) {
    GroupStepWork *w = static_cast<GroupStepWork *>(data);
    w->status = ge_group_step(w->box->g, w->turns);            // all devices step concurrently
    if (w->status == GE_OK) w->status = ge_group_sync(w->box->g);
}
void group_step_complete(napi_env env, napi_status, void *data) {
    GroupStepWork *w = static_cast<GroupStepWork *>(data);
    w->box->busy = false;
    napi_value v;
    if (w->status == GE_OK) {
        napi_get_undefined(env, &v);
        napi_resolve_deferred(env, w->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, ge_strerror(w->status), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &v);
        napi_reject_deferred(env, w->deferred, v);
    }
    napi_delete_async_work(env, w->work);
    napi_delete_reference(env, w->keep);
    delete w;
}

// groupStep(group, nTurns): Promise<void> - on the libuv pool, like step()
napi_value GroupStep(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_group *g = argc >= 1 ? group_arg(env, argv[0]) : nullptr;
    uint64_t turns = 1;
    if (!g || (argc > 1 && !get_u64(env, argv[1], &turns))) return throw_status(env, GE_ERR_ARG, "groupStep");
    GroupStepWork *w = new GroupStepWork();
    w->box = group_box(env, argv[0]); w->turns = (uint32_t)turns; w->status = GE_OK;
    napi_value promise, name;
    if (napi_create_reference(env, argv[0], 1, &w->keep) != napi_ok) { delete w; return throw_status(env, GE_ERR_ARG, "groupStep"); }
    if (napi_create_promise(env, &w->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "ge_group_step", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, nullptr, name, group_step_execute, group_step_complete, w, &w->work) != napi_ok) {
        napi_delete_reference(env, w->keep);
        delete w;
        return throw_status(env, GE_ERR_ARG, "groupStep");
    }
    w->box->busy = true;
    if (napi_queue_async_work(env, w->work) != napi_ok) {
        w->box->busy = false;
        napi_delete_async_work(env, w->work);
        napi_delete_reference(env, w->keep);
        delete w;
        return throw_status(env, GE_ERR_ARG, "groupStep");
    }
    return promise;
}

// groupSummary(group): ArrayBuffer holding the whole job's ge_summary (per-device reductions + ONE ncclAllGather + sum)
napi_value GroupSummary(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_group *g = argc >= 1 ? group_arg(env, argv[0]) : nullptr;
    if (!g) return throw_status(env, GE_ERR_ARG, "groupSummary");
    void *data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_arraybuffer(env, sizeof(ge_summary), &data, &buf));
    int st = ge_group_summary(g, static_cast<ge_summary *>(data));
    if (st != GE_OK) return throw_status(env, st, "groupSummary");
    return buf;
}

// groupReadRooms(group, shard, first, count): ArrayBuffer of ge_room_view from device `shard`'s batch (local indices)
napi_value GroupReadRooms(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ge_group *g = argc >= 1 ? group_arg(env, argv[0]) : nullptr;
    uint64_t shard = 0, first = 0, count = 0;
    if (!g || argc < 4 || !get_u64(env, argv[1], &shard) || !get_u64(env, argv[2], &first) || !get_u64(env, argv[3], &count))
        return throw_status(env, GE_ERR_ARG, "groupReadRooms");
    ge_batch *b = nullptr;
    int st = ge_group_shard(g, (int)shard, &b);
    if (st != GE_OK) return throw_status(env, st, "groupReadRooms");
    void *data = nullptr;
    napi_value buf;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)count * sizeof(ge_room_view), &data, &buf));
    st = ge_batch_read_rooms(b, first, count, static_cast<ge_room_view *>(data), (size_t)count * sizeof(ge_room_view));
    if (st != GE_OK) return throw_status(env, st, "groupReadRooms");
    return buf;
}

napi_value DestroyGroup(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    GroupBox *box = argc >= 1 ? group_box(env, argv[0]) : nullptr;
    if (!box) return throw_status(env, GE_ERR_ARG, "destroyGroup");
    if (box->busy) { napi_throw_error(env, "GE_BUSY", "destroyGroup while an async step() is in flight"); return nullptr; }
    if (box->g) { ge_group_destroy(box->g); box->g = nullptr; }
    return nullptr;
}

napi_value Init(napi_env env, napi_value exports) {
    napi_property_descriptor props[] = {
        {"compileTable", nullptr, CompileTable, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"tableInfo", nullptr, TableInfo, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createBatch", nullptr, CreateBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"step", nullptr, Step, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"stepSync", nullptr, StepSync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"readRooms", nullptr, ReadRooms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"injectAction", nullptr, InjectAction, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"injectActions", nullptr, InjectActions, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"setTurn", nullptr, SetTurn, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"writeRooms", nullptr, WriteRooms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"destroyBatch", nullptr, DestroyBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"readEvents", nullptr, ReadEvents, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"summary", nullptr, Summary, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"reset", nullptr, Reset, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"roomViewSize", nullptr, RoomViewSize, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createGroup", nullptr, CreateGroup, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupStep", nullptr, GroupStep, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupSummary", nullptr, GroupSummary, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupReadRooms", nullptr, GroupReadRooms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"destroyGroup", nullptr, DestroyGroup, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
