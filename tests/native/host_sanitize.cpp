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

int main(int argc, char **argv) {
    int compiled = 0;
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
