// hipemu runtime: fiber scheduler + counting barriers (see hip/hip_runtime.h).  Test-only.
#include <hip/hip_runtime.h>

__attribute__((aligned(16))) char lama_smem[160 * 1024 + 256];

namespace hipemu {

Block* g_blk = nullptr;
static const size_t kStack = 256 * 1024;

void yield() {
    Block* b = g_blk;
    swapcontext(&b->cur->ctx, &b->sched);
}

void block_barrier() {
    Block* b = g_blk;
    int gen = b->gen;
    if (++b->count >= b->alive) {
        b->count = 0;
        b->gen++;
    } else {
        while (b->gen == gen) yield();
    }
}

void wave_barrier() {
    Wave& w = cur_wave();
    int gen = w.gen;
    if (++w.count >= w.alive) {
        w.count = 0;
        w.gen++;
    } else {
        while (w.gen == gen) yield();
    }
}

static void trampoline() {
    Block* b = g_blk;
    Fiber* f = b->cur;
    b->body();
    f->done = true;
    // a thread that exits no longer participates in barriers
    b->alive--;
    Wave& w = b->waves[f->linear / 64];
    w.alive--;
    if (b->alive > 0 && b->count >= b->alive) { b->count = 0; b->gen++; }
    if (w.alive > 0 && w.count >= w.alive) { w.count = 0; w.gen++; }
    swapcontext(&f->ctx, &b->sched);
}

void run_grid(dim3 grid, dim3 block, std::function<void()> body) {
    int nthreads = (int)(block.x * block.y * block.z);
    Block blk;
    blk.bdim = block;
    blk.gdim = grid;
    blk.body = body;
    blk.nthreads = nthreads;
    blk.fibers.resize(nthreads);
    for (auto& f : blk.fibers) f.stack.resize(kStack);
    int nwaves = (nthreads + 63) / 64;
    Block* saved = g_blk;
    g_blk = &blk;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blk.bid = dim3(bx, by, bz);
                blk.alive = nthreads;
                blk.count = 0;
                blk.gen = 0;
                blk.waves.assign(nwaves, Wave());
                for (int w = 0; w < nwaves; ++w) {
                    int n = nthreads - w * 64;
                    blk.waves[w].nlanes = blk.waves[w].alive = n > 64 ? 64 : n;
                }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = blk.fibers[t];
                    f.linear = t;
                    f.done = false;
                    f.xcount = 0;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack.data();
                    f.ctx.uc_stack.ss_size = f.stack.size();
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    remaining = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = blk.fibers[t];
                        if (f.done) continue;
                        blk.cur = &f;
                        swapcontext(&blk.sched, &f.ctx);
                        if (!f.done) remaining++;
                    }
                }
            }
    g_blk = saved;
}

}  // namespace hipemu
