// tests/hipemu/hipemu.cpp -- fiber scheduler behind tests/hipemu/device_rt.h.
// TEST INFRASTRUCTURE ONLY (see the header).  One ucontext fiber per workgroup thread;
// workgroups run sequentially; barriers / wave rendezvous yield to the scheduler.
#include "device_rt.h"
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace hipemu {

Ctx g;

namespace {
enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    ucontext_t uc;
    char* stack = nullptr;
    State st = DONE;
    Ctx ctx;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber*> pool;
ucontext_t sched_uc;
Fiber* cur = nullptr;
const std::function<void()>* body = nullptr;
std::vector<char> smem;
std::vector<char> wscratch;           // per-wave 64*64 bytes
std::vector<unsigned long long> wlive;  // per-wave live-lane mask

void trampoline() {
    (*body)();
    cur->st = DONE;
    swapcontext(&cur->uc, &sched_uc);
}
void yield(State s) {
    Fiber* f = cur;
    f->st = s;
    swapcontext(&f->uc, &sched_uc);
}
}  // namespace

void* dyn_smem() { return smem.data(); }
void block_sync() { yield(WAIT_BLOCK); }
void wave_sync() { yield(WAIT_WAVE); }
void* wave_scratch() { return wscratch.data() + (size_t)g.wave * 64 * 64; }
unsigned long long wave_live_mask() { return wlive[g.wave]; }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn) {
    if (cur != nullptr) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
    const int nthr = (int)(block.x * block.y * block.z);
    const int nwave = (nthr + 63) / 64;
    if (nthr <= 0 || nthr > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nthr); abort(); }
    while ((int)pool.size() < nthr) {
        Fiber* f = new Fiber();
        f->stack = (char*)malloc(kStack);
        pool.push_back(f);
    }
    smem.assign(shmem + 64, 0);
    wscratch.assign((size_t)nwave * 64 * 64, 0);
    wlive.assign(nwave, 0);
    body = &fn;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int w = 0; w < nwave; ++w) wlive[w] = 0;
        for (int t = 0; t < nthr; ++t) {
            Fiber* f = pool[t];
            getcontext(&f->uc);
            f->uc.uc_stack.ss_sp = f->stack;
            f->uc.uc_stack.ss_size = kStack;
            f->uc.uc_link = &sched_uc;
            makecontext(&f->uc, (void (*)())trampoline, 0);
            f->st = READY;
            f->ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f->ctx.bid = dim3(bx, by, bz);
            f->ctx.bdim = block;
            f->ctx.gdim = grid;
            f->ctx.linear = t;
            f->ctx.lane = t & 63;
            f->ctx.wave = t >> 6;
            wlive[t >> 6] |= (1ull << (t & 63));
        }
        int live = nthr;
        while (live > 0) {
            bool ran = false;
            for (int t = 0; t < nthr; ++t) {
                Fiber* f = pool[t];
                if (f->st != READY) continue;
                ran = true;
                cur = f;
                g = f->ctx;
                swapcontext(&sched_uc, &f->uc);
                cur = nullptr;
                if (f->st == DONE) {
                    --live;
                    wlive[t >> 6] &= ~(1ull << (t & 63));
                }
            }
            // release wave rendezvous whose live lanes have all arrived
            bool released = false;
            for (int w = 0; w < nwave; ++w) {
                int lo = w * 64, hi = std::min(nthr, lo + 64);
                int nlive = 0, nwait = 0;
                for (int t = lo; t < hi; ++t) {
                    if (pool[t]->st != DONE) ++nlive;
                    if (pool[t]->st == WAIT_WAVE) ++nwait;
                }
                if (nlive > 0 && nwait == nlive) {
                    for (int t = lo; t < hi; ++t)
                        if (pool[t]->st == WAIT_WAVE) pool[t]->st = READY;
                    released = true;
                }
            }
            // release the workgroup barrier
            int nblock = 0;
            for (int t = 0; t < nthr; ++t)
                if (pool[t]->st == WAIT_BLOCK) ++nblock;
            if (live > 0 && nblock == live) {
                for (int t = 0; t < nthr; ++t)
                    if (pool[t]->st == WAIT_BLOCK) pool[t]->st = READY;
                released = true;
            }
            if (live > 0 && !ran && !released) {
                fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): divergent barrier / wave op\n", bx, by, bz);
                abort();
            }
        }
    }
    body = nullptr;
}

}  // namespace hipemu
