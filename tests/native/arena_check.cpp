/* Host check of csrc/arena.h: the arena's book-keeping over plain memory.  Random churn of blocks that are scribbled on while live:
 *   - a block said to hold zeros holds zeros (with or without a fill),
 *   - live blocks never overlap (every block carries its own byte pattern until it is released),
 *   - blocks of a page or more start on the page alignment,
 *   - entirely free chunks go back (all but one spare and the first), and everything is returned at destroy.
 * usage: arena_check <seed> <operations> [defer]; prints "OK ..." or the first violation (exit 1).
 * defer = 1: the device arena's mode (ctx.h): release() frees no chunk, trim() -- called every 997 operations here -- does. */
#include <cstdio>
#include <cstring>
#include <random>

#include "arena.h"

struct HostBackend {
    using err_t = int;
    using stream_t = int;
    static long outstanding;
    static size_t fill_calls;
    static err_t ok() { return 0; }
    static err_t malloc(void **p, size_t n) {
        if (posix_memalign(p, 2u << 20, n) != 0) return 1;
        memset(*p, 0xEE, n);                                   /* device memory does not arrive as zeros */
        outstanding++;
        return 0;
    }
    static void free(void *p) { ::free(p); outstanding--; }
    static err_t fill_zero(void *p, size_t n, stream_t) { memset(p, 0, n); fill_calls++; return 0; }
    static err_t wait(stream_t) { return 0; }
    static void wait_device() {}
};
long HostBackend::outstanding = 0;
size_t HostBackend::fill_calls = 0;

struct Block { unsigned char *p; size_t n; unsigned char tag; };

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int ops = argc > 2 ? atoi(argv[2]) : 20000;
    std::mt19937 rng(seed);
    ArenaT<HostBackend> arena;
    arena.first_chunk = 64 << 10;
    arena.defer_trim = argc > 3 && atoi(argv[3]) != 0;
    std::vector<Block> blocks;
    size_t peak_chunks = 0;
    auto fail = [&](const char *what, int op) { printf("FAIL seed %u op %d: %s\n", seed, op, what); return 1; };
    for (int op = 0; op < ops; op++) {
        const int phase = (op / 2000) % 3;                     /* grow, churn, shrink */
        const unsigned want_alloc = phase == 0 ? 70 : phase == 1 ? 50 : 25;
        if (blocks.empty() || rng() % 100 < want_alloc) {
            size_t n;
            switch (rng() % 4) {
            case 0: n = 1 + rng() % 600; break;
            case 1: n = 4096 + rng() % 40000; break;
            case 2: n = 100000 + rng() % 300000; break;
            default: n = 256 * (1 + rng() % 64); break;
            }
            void *p = nullptr;
            const bool zero = rng() % 3 != 0;
            if ((zero ? arena.alloc_zeroed(&p, n, 0) : arena.alloc(&p, n)) != 0 || !p) return fail("allocation refused", op);
            unsigned char *b = static_cast<unsigned char *>(p);
            if (n >= 4096 && ((uintptr_t)b & 4095) != 0) return fail("large block not on a page", op);
            if (((uintptr_t)b & 255) != 0) return fail("block not on 256 bytes", op);
            if (zero) for (size_t i = 0; i < n; i++) if (b[i]) return fail("a block of zeros is not zeros", op);
            const unsigned char tag = (unsigned char)(1 + rng() % 254);
            memset(b, tag, n);
            blocks.push_back(Block{ b, n, tag });
        } else {
            const size_t i = rng() % blocks.size();
            Block k = blocks[i];
            blocks[i] = blocks.back();
            blocks.pop_back();
            for (size_t j = 0; j < k.n; j++) if (k.p[j] != k.tag) return fail("a live block was overwritten (overlap)", op);
            arena.release(k.p);
        }
        peak_chunks = std::max(peak_chunks, arena.chunks_held());
        if (arena.live.size() != blocks.size()) return fail("live count", op);
        size_t spare = 0;
        for (size_t c = 1; c < arena.chunks.size(); c++) spare += ArenaT<HostBackend>::entirely_free(arena.chunks[c]);
        if (arena.defer_trim) {
            const long before = HostBackend::outstanding;
            if (op % 997 == 996) {
                arena.trim();
                spare = 0;
                for (size_t c = 1; c < arena.chunks.size(); c++) spare += ArenaT<HostBackend>::entirely_free(arena.chunks[c]);
                if (spare > 1) return fail("more than one spare chunk after trim()", op);
            } else if (HostBackend::outstanding < before) return fail("a deferred arena gave a chunk back outside trim()", op);
        } else if (spare > 1) return fail("more than one spare chunk", op);
    }
    for (auto &k : blocks) {
        for (size_t j = 0; j < k.n; j++) if (k.p[j] != k.tag) return fail("a live block was overwritten (overlap)", ops);
        arena.release(k.p);
    }
    arena.trim();
    if (arena.chunks_held() > 2) return fail("free chunks kept", ops);
    const size_t held = arena.chunks_held();
    arena.destroy();
    if (HostBackend::outstanding != 0) return fail("chunks leaked", ops);
    printf("OK seed %u: %d operations, peak %zu chunks, %zu given back early, %zu held at the end, zero fills issued %zu avoided %zu\n", seed, ops,
           peak_chunks, arena.trimmed, held, arena.fills, arena.fills_saved);
    return 0;
}
