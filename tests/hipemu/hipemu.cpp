// tests/hipemu/hipemu.cpp -- fiber scheduler behind tests/hipemu/device_rt.h.
// TEST INFRASTRUCTURE ONLY (see the header).  One fiber per workgroup thread (hand-rolled x86-64
// context switch, no syscalls); workgroups run sequentially; barriers / wave rendezvous yield to
// the scheduler.
#include "device_rt.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

Ctx g;

namespace {
enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    State st = DONE;
    Ctx ctx;
};
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber*> pool;
void* sched_sp = nullptr;
Fiber* cur = nullptr;
const std::function<void()>* body = nullptr;
std::vector<char> smem;
std::vector<char> wscratch;             // per-wave 2 x 64*64 bytes (ping-pong)
std::vector<unsigned long long> wlive;  // per-wave live-lane mask
std::vector<int> wparity;               // per-wave scratch parity

void trampoline() {
    (*body)();
    cur->st = DONE;
    hipemu_switch(&cur->sp, sched_sp);
    abort();
}
void yield(State s) {
    Fiber* f = cur;
    f->st = s;
    hipemu_switch(&f->sp, sched_sp);
}
void prepare(Fiber* f) {
    uintptr_t top = ((uintptr_t)(f->stack + kStack)) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                // fake return address of trampoline (keeps rsp % 16 == 8 at entry)
    *--sp = (void*)&trampoline;     // `ret` target of the first switch
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    f->sp = (void*)sp;
}
}  // namespace

void* dyn_smem() { return smem.data(); }
void block_sync() { yield(WAIT_BLOCK); }
void wave_sync() { yield(WAIT_WAVE); }
void* wave_scratch() { return wscratch.data() + ((size_t)g.wave * 2 + 0) * 64 * 64; }
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
    wscratch.assign((size_t)nwave * 2 * 64 * 64, 0);
    wlive.assign(nwave, 0);
    body = &fn;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int w = 0; w < nwave; ++w) wlive[w] = 0;
        for (int t = 0; t < nthr; ++t) {
            Fiber* f = pool[t];
            prepare(f);
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
                hipemu_switch(&sched_sp, f->sp);
                cur = nullptr;
                if (f->st == DONE) {
                    --live;
                    wlive[t >> 6] &= ~(1ull << (t & 63));
                }
            }
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
