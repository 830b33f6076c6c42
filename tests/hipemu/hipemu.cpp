// hipemu.cpp -- TEST INFRASTRUCTURE ONLY (see hipemu.h).
#include "hipemu.h"


#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

// minimal x86-64 System V context switch (callee-saved registers + stack pointer); ucontext's swapcontext costs two
// sigprocmask syscalls per switch, which dominated the emulator's run time.
extern "C" void hipemu_ctx_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hipemu_ctx_switch
.type hipemu_ctx_switch,@function
hipemu_ctx_switch:
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
.size hipemu_ctx_switch, .-hipemu_ctx_switch
)");

namespace hipemu {

thread_local ThreadCtx* g_cur = nullptr;

namespace {

constexpr size_t kStack = 256 * 1024;
constexpr size_t kSlot = 64;  // bytes per lane in a wave exchange buffer

struct Fiber {
  void* sp = nullptr;
  ThreadCtx ctx;
  bool done = false;
  char* stack = nullptr;
};

struct WaveState {
  int arrived = 0;
  unsigned gen = 0;
  int alive = 64;
  alignas(16) char buf[64 * kSlot];
};

struct BlockRun {
  void* sched_sp = nullptr;
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  int alive = 0;
  Fiber* running = nullptr;
  const std::function<void()>* body = nullptr;
  unsigned long progress = 0;
};

thread_local BlockRun* t_run = nullptr;

void yield_to_sched() {
  BlockRun* r = t_run;
  Fiber* f = r->running;
  hipemu_ctx_switch(&f->sp, r->sched_sp);
  g_cur = &f->ctx;
}

void fiber_main() {
  BlockRun* r = t_run;
  Fiber* f = r->running;
  g_cur = &f->ctx;
  (*r->body)();
  f->done = true;
  r->alive--;
  r->waves[f->ctx.wave].alive--;
  r->progress++;
  hipemu_ctx_switch(&f->sp, r->sched_sp);
  __builtin_trap();   // a finished fiber is never resumed
}

}  // namespace

void sync_threads() {
  BlockRun* r = t_run;
  r->bar_arrived++;
  r->progress++;
  const unsigned my = r->bar_gen;
  if (r->bar_arrived >= r->alive) {
    r->bar_arrived = 0;
    r->bar_gen++;
    return;
  }
  while (r->bar_gen == my) {
    yield_to_sched();
    if (r->bar_gen == my && r->bar_arrived >= r->alive) {  // someone exited meanwhile
      r->bar_arrived = 0;
      r->bar_gen++;
    }
  }
}

void wave_barrier() {
  BlockRun* r = t_run;
  WaveState& w = r->waves[g_cur->wave];
  w.arrived++;
  r->progress++;
  const unsigned my = w.gen;
  if (w.arrived >= w.alive) {
    w.arrived = 0;
    w.gen++;
    return;
  }
  while (w.gen == my) yield_to_sched();
}

const char* wave_exchange(const void* mine, size_t nbytes, size_t* slot_stride) {
  if (nbytes > kSlot) { fprintf(stderr, "hipemu: exchange payload too large\n"); abort(); }
  BlockRun* r = t_run;
  WaveState& w = r->waves[g_cur->wave];
  std::memcpy(w.buf + (size_t)g_cur->lane * kSlot, mine, nbytes);
  wave_barrier();
  *slot_stride = kSlot;
  return w.buf;
}

void wave_release() { wave_barrier(); }

// fiber stacks are recycled through a process-wide free list: a fresh 256 KB malloc per emulated thread per workgroup is
// an mmap + page faults + munmap each (measured: more system time than user time for the whole CPU test tier)
static std::mutex g_stack_mu;
static std::vector<char*> g_stack_free;

static void acquire_stacks(std::vector<char*>& out, size_t n) {
  out.clear();
  {
    std::lock_guard<std::mutex> lk(g_stack_mu);
    while (out.size() < n && !g_stack_free.empty()) {
      out.push_back(g_stack_free.back());
      g_stack_free.pop_back();
    }
  }
  while (out.size() < n) out.push_back((char*)malloc(kStack));
}

static void release_stacks(std::vector<char*>& v) {
  std::lock_guard<std::mutex> lk(g_stack_mu);
  for (char* p : v) g_stack_free.push_back(p);
  v.clear();
}

static void run_block(const std::function<void()>& body, dim3 grid, dim3 block, unsigned bx, unsigned by,
                      unsigned bz, size_t dyn_smem_bytes) {
  const unsigned nthreads = block.x * block.y * block.z;
  BlockRun run;
  run.body = &body;
  run.fibers.resize(nthreads);
  run.waves.resize((nthreads + 63) / 64);
  run.alive = (int)nthreads;
  std::vector<char> smem(dyn_smem_bytes + 64);
  char* smem_aligned = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
  t_run = &run;
  std::vector<char*> stacks;
  acquire_stacks(stacks, nthreads);
  for (unsigned t = 0; t < nthreads; ++t) {
    Fiber& f = run.fibers[t];
    f.stack = stacks[t];
    f.ctx.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
    f.ctx.bid = {bx, by, bz};
    f.ctx.bdim = {block.x, block.y, block.z};
    f.ctx.gdim = {grid.x, grid.y, grid.z};
    f.ctx.dyn_smem = smem_aligned;
    f.ctx.lane = (int)(t & 63);
    f.ctx.wave = (int)(t >> 6);
    // initial frame: six zeroed callee-saved registers, then the entry address consumed by `ret`
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int q = 0; q < 6; ++q) sp[q] = nullptr;
    sp[6] = (void*)fiber_main;
    sp[7] = nullptr;
    f.sp = (void*)sp;
  }
  for (unsigned w = 0; w < run.waves.size(); ++w) {
    unsigned lo = w * 64, hi = lo + 64 > nthreads ? nthreads : lo + 64;
    run.waves[w].alive = (int)(hi - lo);
  }
  while (run.alive > 0) {
    const unsigned long before = run.progress;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = run.fibers[t];
      if (f.done) continue;
      run.running = &f;
      hipemu_ctx_switch(&run.sched_sp, f.sp);
    }
    if (run.progress == before && run.alive > 0) {
      fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %d threads alive, barrier %d arrived\n", bx, by, bz,
              run.alive, run.bar_arrived);
      abort();
    }
  }
  release_stacks(stacks);
  t_run = nullptr;
  g_cur = nullptr;
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  unsigned nthr = std::thread::hardware_concurrency();
  if (const char* e = getenv("HIPEMU_THREADS")) nthr = (unsigned)atoi(e);
  if (nthr < 1) nthr = 1;
  if (nthr > nblocks) nthr = (unsigned)nblocks;
  std::atomic<unsigned long> next{0};
  auto worker = [&]() {
    for (;;) {
      unsigned long b = next.fetch_add(1);
      if (b >= nblocks) break;
      unsigned bx = (unsigned)(b % grid.x), by = (unsigned)((b / grid.x) % grid.y),
               bz = (unsigned)(b / ((unsigned long)grid.x * grid.y));
      run_block(body, grid, block, bx, by, bz, dyn_smem_bytes);
    }
  };
  if (nthr == 1) { worker(); return; }
  std::vector<std::thread> pool;
  for (unsigned i = 0; i < nthr; ++i) pool.emplace_back(worker);
  for (auto& t : pool) t.join();
}

}  // namespace hipemu
