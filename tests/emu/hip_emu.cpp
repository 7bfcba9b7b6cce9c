// hip_emu.cpp -- runtime of the CPU fiber emulator (see hip_emu.h). TEST INFRASTRUCTURE ONLY.
#include "hip_emu.h"

#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {

thread_local ThreadCtx* tls_cur = nullptr;

namespace {

constexpr size_t kStackBytes = 256 * 1024;

struct WaveState {
  int nlanes = 0;
  int count = 0;
  unsigned gen = 0;
  float xa[2][kWave];
  float xb[2][kWave];
  int xi[2][kWave];
  uint4 qa[2][kWave];
  uint4 qb[2][kWave];
};

struct Fiber {
  ucontext_t ctx;
  ThreadCtx tc;
  bool done = false;
  // blocked while *wait_gen == wait_val
  const unsigned* wait_gen = nullptr;
  unsigned wait_val = 0;
};

struct BlockRunner {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  std::vector<char> stacks;
  std::vector<WaveState> waves;
  std::vector<char> smem;
  int nthreads = 0;
  int block_count = 0;  // arrivals at the block barrier
  unsigned block_gen = 0;
  Fiber* cur = nullptr;
  const std::function<void()>* body = nullptr;
};

thread_local BlockRunner* tls_runner = nullptr;

void fiber_entry() {
  BlockRunner* r = tls_runner;
  Fiber* f = r->cur;
  (*r->body)();
  f->done = true;
  swapcontext(&f->ctx, &r->sched);
}

void yield_wait(const unsigned* gen, unsigned val) {
  BlockRunner* r = tls_runner;
  Fiber* f = r->cur;
  f->wait_gen = gen;
  f->wait_val = val;
  swapcontext(&f->ctx, &r->sched);
}

void run_block(BlockRunner& r, dim3 grid, dim3 block, dim3 bidx, size_t smem_bytes,
               const std::function<void()>& body) {
  const int nt = (int)(block.x * block.y * block.z);
  r.nthreads = nt;
  r.body = &body;
  r.block_count = 0;
  r.block_gen = 0;
  if ((int)r.fibers.size() < nt) r.fibers.resize(nt);
  if (r.stacks.size() < (size_t)nt * kStackBytes) r.stacks.resize((size_t)nt * kStackBytes);
  const int nw = (nt + kWave - 1) / kWave;
  r.waves.assign(nw, WaveState());
  for (int w = 0; w < nw; ++w) r.waves[w].nlanes = std::min(kWave, nt - w * kWave);
  r.smem.assign(smem_bytes + 64, 0);
  for (int t = 0; t < nt; ++t) {
    Fiber& f = r.fibers[t];
    f.done = false;
    f.wait_gen = nullptr;
    f.tc.tid = t;
    f.tc.lane = t % kWave;
    f.tc.wave = t / kWave;
    f.tc.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    f.tc.bidx = bidx;
    f.tc.bdim = block;
    f.tc.gdim = grid;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = r.stacks.data() + (size_t)t * kStackBytes;
    f.ctx.uc_stack.ss_size = kStackBytes;
    f.ctx.uc_link = &r.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  tls_runner = &r;
  int remaining = nt;
  while (remaining > 0) {
    bool progressed = false;
    for (int t = 0; t < nt; ++t) {
      Fiber& f = r.fibers[t];
      if (f.done) continue;
      if (f.wait_gen && *f.wait_gen == f.wait_val) continue;  // still blocked
      f.wait_gen = nullptr;
      r.cur = &f;
      tls_cur = &f.tc;
      swapcontext(&r.sched, &f.ctx);
      progressed = true;
      if (f.done) --remaining;
    }
    if (!progressed) {
      std::fprintf(stderr,
                   "hip_emu: DEADLOCK in block (%u,%u,%u): %d threads blocked at a barrier or wave "
                   "operation that the others never reach (divergent barrier / early return?)\n",
                   bidx.x, bidx.y, bidx.z, remaining);
      std::abort();
    }
  }
  tls_cur = nullptr;
  tls_runner = nullptr;
}

// ---- registered buffers ---------------------------------------------------------------
struct Range {
  uintptr_t lo, hi;
};
std::mutex g_buf_mu;
std::vector<Range> g_bufs;
std::atomic<int> g_strict{0};
std::atomic<int> g_violations{0};

}  // namespace

namespace {
std::mutex g_pool_mu;
std::vector<BlockRunner*> g_pool;
BlockRunner* acquire_runner() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.empty()) return new BlockRunner();
  BlockRunner* r = g_pool.back();
  g_pool.pop_back();
  return r;
}
void release_runner(BlockRunner* r) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back(r);
}
}  // namespace

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  if (nblocks <= 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("ECO_EMU_THREADS")) hw = (unsigned)std::atoi(e);
  const int nworkers = (int)std::max<long>(1, std::min<long>(hw ? hw : 1, nblocks));
  std::atomic<long> next{0};
  auto worker = [&]() {
    BlockRunner* rp = acquire_runner();
    BlockRunner& runner = *rp;
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= nblocks) break;
      dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(runner, grid, block, bidx, dyn_smem_bytes, body);
    }
    release_runner(rp);
  };
  if (nworkers == 1) {
    worker();
    return;
  }
  std::vector<std::thread> th;
  for (int i = 0; i < nworkers; ++i) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

void syncthreads() {
  BlockRunner* r = tls_runner;
  const unsigned g = r->block_gen;
  if (++r->block_count == r->nthreads) {
    r->block_count = 0;
    r->block_gen = g + 1;
    return;
  }
  yield_wait(&r->block_gen, g);
}

static inline WaveState& my_wave() { return tls_runner->waves[tls_cur->wave]; }

// Rendezvous of all lanes of the calling wave; returns the exchange slot used for this op.
static inline int wave_rendezvous_begin(WaveState& w) { return (int)(w.gen & 1u); }
static inline void wave_rendezvous_wait(WaveState& w) {
  const unsigned g = w.gen;
  if (++w.count == w.nlanes) {
    w.count = 0;
    w.gen = g + 1;
    return;
  }
  yield_wait(&w.gen, g);
}

float wave_xchg_f32(float v, int src_lane) {
  WaveState& w = my_wave();
  const int slot = wave_rendezvous_begin(w);
  w.xa[slot][tls_cur->lane] = v;
  wave_rendezvous_wait(w);
  if (src_lane < 0 || src_lane >= w.nlanes) return v;  // hardware: own value for out-of-range source
  return w.xa[slot][src_lane];
}

int readfirstlane(int v) {
  WaveState& w = my_wave();
  const int slot = wave_rendezvous_begin(w);
  w.xi[slot][tls_cur->lane] = v;
  wave_rendezvous_wait(w);
  return w.xi[slot][0];
}

emu_f32x16 mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
  WaveState& w = my_wave();
  if (w.nlanes != kWave) {
    std::fprintf(stderr, "hip_emu: MFMA issued by a partial wave (%d lanes)\n", w.nlanes);
    std::abort();
  }
  const int slot = wave_rendezvous_begin(w);
  const int lane = tls_cur->lane;
  w.xa[slot][lane] = a;
  w.xb[slot][lane] = b;
  wave_rendezvous_wait(w);
  const int j = lane & 31;
  const int hi = lane >> 5;
  emu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = d[r];
    acc = std::fmaf(w.xa[slot][i], w.xb[slot][j], acc);            // k = 0
    acc = std::fmaf(w.xa[slot][i + 32], w.xb[slot][j + 32], acc);  // k = 1
    d[r] = acc;
  }
  return d;
}

static inline float bf16_bits_to_f32(unsigned h) {
  unsigned u = h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

emu_f32x16 mfma_f32_32x32x16_bf16(uint4 a, uint4 b, emu_f32x16 c) {
  WaveState& w = my_wave();
  if (w.nlanes != kWave) {
    std::fprintf(stderr, "hip_emu: MFMA issued by a partial wave (%d lanes)\n", w.nlanes);
    std::abort();
  }
  const int slot = wave_rendezvous_begin(w);
  const int lane = tls_cur->lane;
  w.qa[slot][lane] = a;
  w.qb[slot][lane] = b;
  wave_rendezvous_wait(w);
  const int j = lane & 31;
  const int hi = lane >> 5;
  emu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = d[r];
    for (int g = 0; g < 2; ++g) {          // k = 8*g + e held by lanes i + 32*g (A) and j + 32*g (B)
      const unsigned* pa = &w.qa[slot][i + 32 * g].x;
      const unsigned* pb = &w.qb[slot][j + 32 * g].x;
      for (int e = 0; e < 8; ++e) {
        const unsigned ha = (pa[e >> 1] >> (16 * (e & 1))) & 0xffffu;
        const unsigned hb = (pb[e >> 1] >> (16 * (e & 1))) & 0xffffu;
        acc = std::fmaf(bf16_bits_to_f32(ha), bf16_bits_to_f32(hb), acc);
      }
    }
    d[r] = acc;
  }
  return d;
}

void* dyn_smem() {
  BlockRunner* r = tls_runner;
  uintptr_t p = (uintptr_t)r->smem.data();
  return (void*)((p + 15) & ~(uintptr_t)15);
}

bool check_access(const void* p, size_t bytes, bool write) {
  if (!g_strict.load(std::memory_order_relaxed)) return true;
  const uintptr_t a = (uintptr_t)p;
  {
    std::lock_guard<std::mutex> lk(g_buf_mu);
    for (const Range& r : g_bufs)
      if (a >= r.lo && a + bytes <= r.hi) return true;
  }
  if (g_violations.fetch_add(1) < 8) {
    const ThreadCtx* t = tls_cur;
    std::fprintf(stderr, "hip_emu: OUT-OF-BOUNDS %s of %zu bytes at %p (block %u, thread %d)\n",
                 write ? "store" : "load", bytes, p, t ? t->bidx.x : 0u, t ? t->tid : -1);
  }
  return false;
}

}  // namespace emu

extern "C" {
void emu_register_buffer(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(emu::g_buf_mu);
  emu::g_bufs.push_back(emu::Range{(uintptr_t)p, (uintptr_t)p + bytes});
}
void emu_clear_buffers(void) {
  std::lock_guard<std::mutex> lk(emu::g_buf_mu);
  emu::g_bufs.clear();
  emu::g_violations.store(0);
}
void emu_set_strict(int on) { emu::g_strict.store(on); }
int emu_violation_count(void) { return emu::g_violations.load(); }
}
