// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#include <atomic>
#include <thread>
#include <vector>

namespace cbim_emu {

static const size_t kStack = 96 * 1024;
static const int kMaxThreads = 1024;

struct Fiber {
  ucontext_t uc;
  Lane lane;
  bool done;
  unsigned char* stack;
};

struct BlockCtx {
  ucontext_t sched;
  Fiber* fibers = nullptr;
  int nthreads = 0;
  int cur = 0;
  dim3 bidx, bdim, gdim;
  // block barrier
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  int n_done = 0;
  // wave exchange
  struct WaveX {
    unsigned char* buf[2];
    int arrived = 0;
    unsigned gen = 0;
  } wx[kMaxThreads / 64];
  unsigned char* smem = nullptr;
  size_t smem_cap = 0;
  const std::function<void()>* body = nullptr;
};

static thread_local BlockCtx* g_ctx = nullptr;

BlockCtx* ctx() { return g_ctx; }
Lane* lane() { return &g_ctx->fibers[g_ctx->cur].lane; }
const dim3& block_idx() { return g_ctx->bidx; }
const dim3& block_dim() { return g_ctx->bdim; }
const dim3& grid_dim() { return g_ctx->gdim; }
unsigned char* dyn_smem() { return g_ctx->smem; }

static inline void yield() {
  BlockCtx* c = g_ctx;
  swapcontext(&c->fibers[c->cur].uc, &c->sched);
}

void sync_block() {
  BlockCtx* c = g_ctx;
  unsigned gen = c->bar_gen;
  c->bar_arrived++;
  while (true) {
    if (c->bar_gen != gen) return;
    if (c->bar_arrived + c->n_done >= c->nthreads) {
      c->bar_arrived = 0;
      c->bar_gen++;
      return;
    }
    yield();
  }
}

static const size_t kSlot = 64;  // max bytes per lane per collective

const unsigned char* wave_exchange(const void* mine, size_t bytes) {
  BlockCtx* c = g_ctx;
  if (bytes > kSlot) { fprintf(stderr, "emu: wave_exchange payload too large\n"); abort(); }
  int flat = c->fibers[c->cur].lane.flat;
  int w = flat >> 6, l = flat & 63;
  BlockCtx::WaveX& x = c->wx[w];
  unsigned gen = x.gen;
  unsigned char* buf = x.buf[gen & 1];
  memcpy(buf + (size_t)l * bytes, mine, bytes);
  x.arrived++;
  int wave_lanes = c->nthreads - w * 64;
  if (wave_lanes > 64) wave_lanes = 64;
  while (true) {
    if (x.gen != gen) break;
    if (x.arrived >= wave_lanes) {  // exited lanes would deadlock: kernels keep collectives uniform
      x.arrived = 0;
      x.gen++;
      break;
    }
    yield();
  }
  return buf;
}

static void fiber_main() {
  BlockCtx* c = g_ctx;
  (*c->body)();
  c = g_ctx;
  c->fibers[c->cur].done = true;
  c->n_done++;
  // a finished thread releases anybody waiting on the block barrier
  swapcontext(&c->fibers[c->cur].uc, &c->sched);
}

static void run_block(BlockCtx* c) {
  int n = c->nthreads;
  c->bar_arrived = 0;
  c->n_done = 0;
  for (int w = 0; w < (n + 63) / 64; ++w) { c->wx[w].arrived = 0; }
  for (int t = 0; t < n; ++t) {
    Fiber& f = c->fibers[t];
    f.done = false;
    f.lane.flat = t;
    f.lane.tid = dim3(t % c->bdim.x, (t / c->bdim.x) % c->bdim.y, t / (c->bdim.x * c->bdim.y));
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = f.stack;
    f.uc.uc_stack.ss_size = kStack;
    f.uc.uc_link = &c->sched;
    makecontext(&f.uc, (void (*)())fiber_main, 0);
  }
  int remaining = n;
  long spins = 0;
  while (remaining > 0) {
    remaining = 0;
    for (int t = 0; t < n; ++t) {
      Fiber& f = c->fibers[t];
      if (f.done) continue;
      c->cur = t;
      swapcontext(&c->sched, &f.uc);
      if (!f.done) remaining++;
    }
    if (++spins > 50000000L) { fprintf(stderr, "emu: deadlock suspected in block\n"); abort(); }
  }
}

int g_last_launch_err = 0;
int last_launch_err() { return g_last_launch_err; }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  // like the hardware: a workgroup cannot have more than 160 KiB of LDS (gfx950) -> "invalid argument", nothing runs
  g_last_launch_err = shmem > 160 * 1024 ? 1 : 0;
  if (g_last_launch_err) return;
  int nthreads = block.x * block.y * block.z;
  if (nthreads > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
  long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  int nworkers = (int)std::thread::hardware_concurrency();
  const char* env = getenv("CBIM_EMU_THREADS");
  if (env) nworkers = atoi(env);
  if (nworkers < 1) nworkers = 1;
  if (nworkers > nblocks) nworkers = (int)nblocks;
  std::atomic<long> next(0);
  auto worker = [&]() {
    BlockCtx* c = new BlockCtx();
    c->fibers = new Fiber[nthreads];
    for (int t = 0; t < nthreads; ++t) c->fibers[t].stack = (unsigned char*)malloc(kStack);
    for (int w = 0; w < (nthreads + 63) / 64; ++w) {
      c->wx[w].buf[0] = (unsigned char*)malloc(64 * kSlot);
      c->wx[w].buf[1] = (unsigned char*)malloc(64 * kSlot);
    }
    c->smem = (unsigned char*)aligned_alloc(64, ((shmem + 63) / 64 + 1) * 64);
    c->nthreads = nthreads;
    c->bdim = block;
    c->gdim = grid;
    c->body = &body;
    g_ctx = c;
    while (true) {
      long b = next.fetch_add(1);
      if (b >= nblocks) break;
      c->bidx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(c);
    }
    g_ctx = nullptr;
    for (int t = 0; t < nthreads; ++t) free(c->fibers[t].stack);
    for (int w = 0; w < (nthreads + 63) / 64; ++w) { free(c->wx[w].buf[0]); free(c->wx[w].buf[1]); }
    free(c->smem);
    delete[] c->fibers;
    delete c;
  };
  if (nworkers == 1) {
    worker();
  } else {
    std::vector<std::thread> ths;
    for (int i = 0; i < nworkers; ++i) ths.emplace_back(worker);
    for (auto& t : ths) t.join();
  }
}

}  // namespace cbim_emu
