// Host-only code of libge_step.so under AddressSanitizer + UBSan (CPU; sanitizers cannot run on the GPU pool): the DSL
// compiler (ge_table.cpp) and everything ge_host.h holds - table rows, literal image, restart template, room view <-> packed
// record conversion, write validation.  Driven by tests/test_host_sanitizers.py:   host_sanitize <dsl.json>...
// Exit 0 = every DSL compiled, every conversion round-tripped, nothing the sanitizers object to.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../game_engine_amd/csrc/ge_host.h"

using namespace ge;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static int fail(const char *what, const char *file) { fprintf(stderr, "FAILED: %s (%s)\n", what, file); return 1; }

static ge_room_view random_view(const ge_game_table &tb, uint32_t n, bool valid) {
    ge_room_view v;
    memset(&v, 0, sizeof v);
    v.pack = (uint8_t)tb.pack; v.n_players = (uint8_t)n;
    v.phase_id = tb.rows[rnd() % tb.n_phases].phase_id; v.prev_phase_id = tb.rows[rnd() % tb.n_phases].phase_id;
    v.phase0_done = rnd() & 1; v.end_turn = (rnd() & 3) ? -1 : (int32_t)(rnd() % 65535); v.games = (int32_t)(rnd() % 65536);
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *f = v.players[i];
        if (tb.pack == GE_PACK_WEREWOLF) {
            f[0] = (uint8_t)(rnd() % 5); f[1] = (uint8_t)(rnd() % 3);
            for (int k = 2; k <= 7; k++) f[k] = rnd() & 1;
            f[8] = (uint8_t)(rnd() % (n + 1)); f[9] = rnd() & 1; f[10] = (uint8_t)(rnd() % (n + 1));
            v.det[i] = (uint8_t)(rnd() % 3);
        } else {
            f[0] = rnd() & 1; f[1] = rnd() & 1; f[2] = (uint8_t)(rnd() & 3); f[3] = rnd() & 1; f[4] = rnd() & 1; f[5] = (uint8_t)(rnd() & 3);
            f[6] = rnd() & 1; f[7] = (uint8_t)rnd(); f[8] = (uint8_t)(rnd() & 15); f[9] = rnd() & 1; f[10] = (uint8_t)(rnd() & 3);
        }
    }
    if (!valid) {                                               // what ge_batch_write_rooms must refuse, or must at least survive
        switch (rnd() % 5) {
        case 0: v.n_players = (uint8_t)(rnd() % 20); break;
        case 1: v.pack = (uint8_t)(rnd() % 4); break;
        case 2: v.phase_id = (int32_t)rnd(); break;
        case 3: v.prev_phase_id = -(int32_t)(rnd() % 1000) - 1; break;
        default: v.players[rnd() % n][0] = (uint8_t)(5 + rnd() % 250); break;
        }
    }
    return v;
}

// group_partition (what ge_group_create and ge_group_partition shard a job with): for every n the parts tile every segment
// exactly, in order, and each part's seg_first is the global index those rooms have in ONE batch of the job.
static int check_partition(const ge_game_table *tb_ww, const ge_game_table *tb_tt) {
    struct Job { uint32_t n_seg; uint64_t rooms[GE_MAX_SEGMENTS]; uint64_t first_room; };
    const Job jobs[] = {
        {1, {16777216ull, 0, 0, 0}, 0},                                   // C4: 16 777 216 Werewolf x 12
        {2, {8388608ull, 8388608ull, 0, 0}, 1ull << 40},                   // C5: half Werewolf x 8, half Two-Truths x 4
        {4, {1000003ull, 64ull, 777ull, 4099ull}, 0xFFFFFFFF00ull},        // four ragged segments, one as small as n
        {3, {65ull, 66ull, 67ull, 0}, 5},
    };
    for (const Job &j : jobs) {
        ge_batch_desc d;
        memset(&d, 0, sizeof d);
        d.seed = 7; d.first_room = j.first_room; d.n_segments = j.n_seg; d.flags = GE_FLAG_RESTART; d.max_fuse = 3;
        for (uint32_t k = 0; k < j.n_seg; k++) { d.seg[k].table = (k & 1) ? tb_tt : tb_ww; d.seg[k].n_players = (k & 1) ? 4 : 8; d.seg[k].n_rooms = j.rooms[k]; d.seg[k].human_mask = k; }
        for (int n : {1, 2, 3, 5, 8, 64}) {
            uint64_t next[GE_MAX_SEGMENTS], base[GE_MAX_SEGMENTS], acc = j.first_room;
            for (uint32_t k = 0; k < j.n_seg; k++) { base[k] = next[k] = acc; acc += j.rooms[k]; }
            for (int i = 0; i < n; i++) {
                ge_batch_desc sh;
                uint64_t first[GE_MAX_SEGMENTS];
                if (group_partition(d, n, i, &sh, first) != GE_OK) return 1;
                if (sh.seed != d.seed || sh.first_room != d.first_room || sh.n_segments != d.n_segments || sh.flags != d.flags || sh.max_fuse != d.max_fuse) return 2;
                for (uint32_t k = 0; k < j.n_seg; k++) {
                    if (first[k] != next[k]) return 3;                            // parts follow each other without gap or overlap
                    if (sh.seg[k].table != d.seg[k].table || sh.seg[k].n_players != d.seg[k].n_players || sh.seg[k].human_mask != d.seg[k].human_mask) return 4;
                    const uint64_t want = j.rooms[k] * (uint64_t)(i + 1) / (uint64_t)n - j.rooms[k] * (uint64_t)i / (uint64_t)n;   // floor(R (i+1) / n) - floor(R i / n)
                    if (sh.seg[k].n_rooms != want || want == 0) return 5;
                    next[k] += sh.seg[k].n_rooms;
                }
            }
            for (uint32_t k = 0; k < j.n_seg; k++)
                if (next[k] != base[k] + j.rooms[k]) return 6;                    // ... and end where the segment ends
        }
        // refused, nothing written: more parts than the smallest segment has rooms, bad part index, null outputs
        ge_batch_desc sh; uint64_t first[GE_MAX_SEGMENTS];
        if (j.n_seg == 4 && group_partition(d, 65, 0, &sh, first) != GE_ERR_ARG) return 7;
        if (group_partition(d, 2, 2, &sh, first) != GE_ERR_ARG || group_partition(d, 0, 0, &sh, first) != GE_ERR_ARG ||
            group_partition(d, 2, -1, &sh, first) != GE_ERR_ARG || group_partition(d, 2, 0, nullptr, first) != GE_ERR_ARG ||
            group_partition(d, 2, 0, &sh, nullptr) != GE_ERR_ARG) return 8;
    }
    // rooms near 2^64 / n: the split product stays exact
    ge_batch_desc d; memset(&d, 0, sizeof d);
    d.n_segments = 1; d.seg[0].table = tb_ww; d.seg[0].n_players = 8; d.seg[0].n_rooms = 0xFFFFFFFFFFFFFFF0ull;
    uint64_t total = 0;
    for (int i = 0; i < 7; i++) {
        ge_batch_desc sh; uint64_t first[GE_MAX_SEGMENTS];
        if (group_partition(d, 7, i, &sh, first) != GE_OK || first[0] != total) return 9;
        total += sh.seg[0].n_rooms;
    }
    return total == d.seg[0].n_rooms ? 0 : 10;
}

int main(int argc, char **argv) {
    int compiled = 0;
    {
        ge_game_table a, b;                                  // group_partition only carries the table pointers along
        memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
        const int pc = check_partition(&a, &b);
        if (pc) { fprintf(stderr, "FAILED: group_partition check %d\n", pc); return 1; }
    }
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return fail("open", argv[a]);
        fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
        std::string text((size_t)len, '\0');
        if (fread(&text[0], 1, (size_t)len, f) != (size_t)len) return fail("read", argv[a]);
        fclose(f);
        for (int rounds = 1; rounds <= 3; rounds += 2) {
            ge_game_table tb;
            char err[256];
            if (ge_table_compile_json(text.data(), text.size(), rounds, &tb, err, sizeof err) != GE_OK) return fail(err, argv[a]);
            compiled++;
            const bool ww = tb.pack == GE_PACK_WEREWOLF;
            for (uint32_t n = (ww ? 4u : 3u); n <= 12u; n++) {
                if ((int)n < tb.min_players) continue;
                const uint32_t kind = ww ? (n <= 8 ? K_WW8 : K_WW12) : (n <= 4 ? K_TT4 : n <= 8 ? K_TT8 : K_TT12);
                // the table as the kernels get it
                std::vector<DevTable> dtv(1);
                DevTable &dt = dtv[0];
                memset(&dt, 0, sizeof dt);
                for (int r = 0; r < tb.n_phases; r++) { dt.rows[r] = to_dev_row(tb, tb.rows[r], kind); dt.conds[r] = to_dev_cond(tb.rows[r]); }
                build_cond_image(tb, kind, dt);
                if (dt.cond_n16 * 16u > COND_IMG_BYTES) return fail("literal image overflows", argv[a]);
                const uint32_t ncl = dt.cond_shape & 7u, ln = (dt.cond_shape >> 4) & 7u;
                for (int r = 0; r < tb.n_phases; r++)
                    if (tb.rows[r].generic) {
                        const uint32_t slot = (dt.rows[r].r0 >> ROW_COND_SLOT_SHIFT) & 31u, stride = kind == K_WW12 ? 32u : 16u;
                        if ((slot + 1u) * ncl * ln * stride > dt.cond_n16 * 16u) return fail("row's literals outside the image", argv[a]);
                        if ((dt.rows[r].r0 >> 8) & 7u) return fail("generic row kept its terms", argv[a]);
                    }
                // the restart template
                ge_room_view v0;
                memset(&v0, 0, sizeof v0);
                v0.end_turn = -1; v0.n_players = (uint8_t)n; v0.pack = (uint8_t)tb.pack;
                for (uint32_t i = 0; i < n; i++) memcpy(v0.players[i], tb.init_fields, 12);
                uint32_t w[12] = {0}, regs[20] = {0};
                view_to_words(kind, v0, tb, w);
                init_regs_of(kind, w, regs);
                // canonical views: view -> words -> view is the identity on valid views, and words -> view -> words on what that produced
                for (int it = 0; it < 400; it++) {
                    const ge_room_view v = random_view(tb, n, true);
                    if (!view_fits(v, tb, n)) return fail("a valid view was refused", argv[a]);
                    uint32_t pw[12] = {0}, pw2[12] = {0};
                    ge_room_view back, back2;
                    view_to_words(kind, v, tb, pw);
                    words_to_view(kind, pw, tb, (int)n, back);
                    // (team / role of a werewolf view are stored as given; a Two-Truths view likewise; prev's effect is derived)
                    if (memcmp(&v, &back, sizeof v) != 0) return fail("view -> words -> view differs", argv[a]);
                    view_to_words(kind, back, tb, pw2);
                    words_to_view(kind, pw2, tb, (int)n, back2);
                    if (memcmp(pw, pw2, sizeof pw) != 0 || memcmp(&back, &back2, sizeof back) != 0) return fail("words -> view -> words differs", argv[a]);
                    const ge_room_view bad = random_view(tb, n, false);
                    if (view_fits(bad, tb, n)) {                   // (a random corruption can land on a valid value again)
                        uint32_t bw[12];
                        view_to_words(kind, bad, tb, bw);
                    }
                }
            }
        }
    }
    printf("compiled %d\n", compiled);
    return 0;
}
