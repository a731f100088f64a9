// TEST INFRASTRUCTURE -- never part of the product (see hip/hip_runtime.h in this directory).
// Fiber scheduler behind the host-side HIP stand-in: a work-group is a set of fibers (own stacks, cooperative switch) on one OS thread,
// barriers are cooperative yields, several OS threads each take whole work-groups.
#include "hip/hip_runtime.h"


#include <atomic>
#include <thread>
#include <vector>

namespace hipemu {
namespace {

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr size_t LDS_BYTES = 160 * 1024;
constexpr int MAX_GATHER = 64;            // bytes a lane may deposit in a wave collective

enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };

struct Wave {
    alignas(16) unsigned char box[2][64][MAX_GATHER];
    int live = 0, waiting = 0;
};

// Minimal x86-64 (System V) cooperative context switch: callee-saved registers + stack pointer.  ucontext's
// swapcontext makes two sigprocmask system calls per switch, which dominated the run time.
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch, @function
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
    .size hipemu_switch, .-hipemu_switch
)");

struct Fiber {
    void *sp = nullptr;
    Ctx ctx;
    State state = DONE;
    unsigned seq = 0;                     // collectives this lane has taken part in (selects the mailbox half)
    int wave = 0;
};

struct Worker {                           // one per OS thread
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    unsigned char *stacks = nullptr;     // malloc'ed, never zero-filled: pages are committed as fibers touch them
    unsigned char *lds = nullptr;
    void *sched_sp = nullptr;
    Fiber *cur = nullptr;
    const std::function<void()> *body = nullptr;
    int live = 0, waiting_block = 0;
};

thread_local Worker *tl_worker = nullptr;

void trampoline() {
    Worker *w = tl_worker;
    (*w->body)();
    Fiber *f = w->cur;
    f->state = DONE;
    w->live--;
    w->waves[f->wave].live--;
    hipemu_switch(&f->sp, w->sched_sp);
    abort();      // a finished fiber is never resumed
}

void yield_to_scheduler() {
    Worker *w = tl_worker;
    hipemu_switch(&w->cur->sp, w->sched_sp);
}

void run_block(Worker *w, Idx block, dim3 bdim, dim3 gdim) {
    const int n = (int)bdim.x;
    const int nwaves = (n + 63) / 64;
    for (int wv = 0; wv < nwaves; ++wv) { w->waves[wv].live = 0; w->waves[wv].waiting = 0; }
    for (int t = 0; t < n; ++t) {
        Fiber &f = w->fibers[t];
        // fresh stack: six zeroed callee-saved slots, then the entry point as return address; the ABI wants
        // rsp % 16 == 8 at function entry (as after a call), hence the extra slot above it
        void **top = (void **)(w->stacks + (size_t)(t + 1) * STACK_BYTES);
        top -= 2;
        top[0] = (void *)trampoline;
        top[1] = nullptr;
        top -= 6;
        for (int i = 0; i < 6; ++i) top[i] = nullptr;
        f.sp = top;
        f.ctx.thread = Idx{(unsigned)t, 0, 0};
        f.ctx.block = block;
        f.ctx.bdim = bdim;
        f.ctx.gdim = gdim;
        f.state = RUNNABLE;
        f.seq = 0;
        f.wave = t >> 6;
        w->waves[f.wave].live++;
    }
    w->live = n;
    w->waiting_block = 0;
    while (w->live > 0) {
        bool progressed = false;
        for (int t = 0; t < n; ++t) {
            Fiber &f = w->fibers[t];
            if (f.state != RUNNABLE) continue;
            w->cur = &f;
            hipemu_switch(&w->sched_sp, f.sp);
            progressed = true;
        }
        bool released = false;
        if (w->live > 0 && w->waiting_block == w->live) {       // every live work-item is at the barrier
            for (int t = 0; t < n; ++t)
                if (w->fibers[t].state == WAIT_BLOCK) w->fibers[t].state = RUNNABLE;
            w->waiting_block = 0;
            released = true;
        }
        for (int wv = 0; wv < nwaves; ++wv) {
            Wave &wave = w->waves[wv];
            if (wave.live > 0 && wave.waiting == wave.live) {
                for (int t = wv * 64; t < std::min(n, wv * 64 + 64); ++t)
                    if (w->fibers[t].state == WAIT_WAVE) w->fibers[t].state = RUNNABLE;
                wave.waiting = 0;
                released = true;
            }
        }
        if (!progressed && !released) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d live, %d at the block barrier\n", block.x, block.y,
                    block.z, w->live, w->waiting_block);
            abort();
        }
    }
}

}  // namespace

Ctx &ctx() { return tl_worker->cur->ctx; }
void *dynamic_shared() { return tl_worker->lds; }

void sync_block() {
    Worker *w = tl_worker;
    w->cur->state = WAIT_BLOCK;
    w->waiting_block++;
    yield_to_scheduler();
}

const unsigned char *wave_gather(const void *mine, size_t bytes) {
    Worker *w = tl_worker;
    Fiber *f = w->cur;
    if (bytes > (size_t)MAX_GATHER) { fprintf(stderr, "hipemu: wave_gather of %zu bytes\n", bytes); abort(); }
    Wave &wave = w->waves[f->wave];
    const unsigned half = f->seq++ & 1;
    memcpy(&wave.box[half][0][0] + (f->ctx.thread.x & 63) * bytes, mine, bytes);     // deposits are packed: slot = lane * bytes
    f->state = WAIT_WAVE;
    wave.waiting++;
    yield_to_scheduler();
    return &wave.box[half][0][0];
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()> &body) {
    if (block.y != 1 || block.z != 1 || block.x == 0 || block.x > 1024 || lds_bytes > LDS_BYTES) {
        fprintf(stderr, "hipemu: unsupported launch (block %u x %u x %u, %zu bytes of LDS)\n", block.x, block.y, block.z, lds_bytes);
        abort();
    }
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char *e = getenv("HIPEMU_THREADS")) nthreads = (unsigned)atoi(e);
    nthreads = (unsigned)std::max<size_t>(1, std::min<size_t>(nthreads ? nthreads : 1, nblocks));
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker w;
        w.fibers.resize(block.x);
        w.waves.resize((block.x + 63) / 64);
        w.stacks = (unsigned char *)malloc((size_t)block.x * STACK_BYTES);
        w.lds = (unsigned char *)aligned_alloc(256, LDS_BYTES);
        w.body = &body;
        tl_worker = &w;
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            Idx idx{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))};
            run_block(&w, idx, block, grid);
        }
        tl_worker = nullptr;
        free(w.lds);
        free(w.stacks);
    };
    if (nthreads == 1) { work(); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(work);
    for (auto &t : pool) t.join();
}

}  // namespace hipemu
