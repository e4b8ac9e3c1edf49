// TEST INFRASTRUCTURE ONLY -- runtime of the host SIMT emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <pthread.h>

#include <algorithm>
#include <cstring>

namespace ha { alignas(16) float smem[40960]; }   // 160 KiB "LDS", one block resident at a time

namespace simt_emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx = nullptr;

// ---- resident teams ---------------------------------------------------------------------------------------------------------------------
// The persistent roll-out kernels need the 32 blocks of an XCD team resident AT THE SAME TIME (they hand activations to one another through
// tagged granules and spin on them).  With g_resident_blocks = n the next launches run all their blocks (<= n) concurrently: one OS thread per
// work-item of EVERY block, every block with its own context and its own NaN-filled LDS (block_lds()).
int g_resident_blocks = 0;
thread_local float* t_lds = nullptr;
float* block_lds() { return t_lds ? t_lds : ha::smem; }

namespace {
struct ResidentArg {
  const std::function<void()>* body;
  BlockCtx* ctx;
  float* lds;
  dim3 grid, block;
  unsigned block_index, thread_index;
};
void* resident_thread(void* p) {
  ResidentArg* a = static_cast<ResidentArg*>(p);
  t_ctx = a->ctx;
  t_lds = a->lds;
  t_blockDim = a->block;
  t_gridDim = a->grid;
  t_threadIdx = dim3(a->thread_index, 0, 0);
  t_blockIdx = dim3(a->block_index, 0, 0);
  (*a->body)();
  return nullptr;
}
}  // namespace

static void launch_resident(dim3 grid, dim3 block, const std::function<void()>& body) {
  const unsigned nthreads = block.x, nwaves = nthreads / 64, nblocks = grid.x;
  if (grid.y != 1 || grid.z != 1 || (int)nblocks > g_resident_blocks) {
    std::fprintf(stderr, "simt_emu: a resident launch takes a 1-D grid of at most g_resident_blocks blocks\n");
    std::abort();
  }
  const unsigned nanbits = 0x7fc00000u;
  float nanv;
  std::memcpy(&nanv, &nanbits, 4);
  std::vector<std::unique_ptr<BlockCtx>> ctxs;
  std::vector<std::unique_ptr<std::barrier<>>> bars;
  std::vector<std::vector<float>> lds(nblocks);
  for (unsigned b = 0; b < nblocks; ++b) {
    ctxs.emplace_back(new BlockCtx);
    bars.emplace_back(new std::barrier<>(nthreads));
    ctxs[b]->block_bar = bars[b].get();
    for (unsigned w = 0; w < nwaves; ++w) {
      ctxs[b]->wave_bar.emplace_back(new std::barrier<>(64));
      ctxs[b]->xch.emplace_back(128, 0);
    }
    lds[b].assign(40960 + 4, nanv);          // (every block starts on NaN-filled LDS, like launch())
  }
  std::vector<ResidentArg> args((size_t)nblocks * nthreads);
  std::vector<pthread_t> tids(args.size());
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, 2u << 20);
  for (unsigned b = 0; b < nblocks; ++b)
    for (unsigned t = 0; t < nthreads; ++t) {
      ResidentArg& a = args[(size_t)b * nthreads + t];
      float* base = lds[b].data();
      while (reinterpret_cast<uintptr_t>(base) & 15) ++base;             // 16-byte aligned, as the kernels' vector accesses expect
      a = ResidentArg{&body, ctxs[b].get(), base, grid, block, b, t};
      if (pthread_create(&tids[(size_t)b * nthreads + t], &attr, resident_thread, &a) != 0) {
        std::fprintf(stderr, "simt_emu: pthread_create failed at block %u thread %u\n", b, t);
        std::abort();
      }
    }
  for (pthread_t& t : tids) pthread_join(t, nullptr);
  pthread_attr_destroy(&attr);
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  if (g_resident_blocks > 0 && block.y == 1 && block.z == 1 && block.x % 64 == 0) {
    launch_resident(grid, block, body);
    return;
  }
  const unsigned nthreads = block.x * block.y * block.z;
  const unsigned nwaves = (nthreads + 63) / 64;
  if (block.y != 1 || block.z != 1 || nthreads % 64 != 0) {
    std::fprintf(stderr, "simt_emu: only 1-D blocks of whole waves are supported\n");
    std::abort();
  }
  // LDS is not cleared between kernels on the hardware, and a kernel that reads a word it never wrote sees whatever the previous tenant of the
  // CU left there (round 5: NaN gradients on one box, green on another).  Every emulated launch therefore starts on NaN-filled "LDS".
  {
    const unsigned nanbits = 0x7fc00000u;
    float nanv;
    std::memcpy(&nanv, &nanbits, 4);
    std::fill(ha::smem, ha::smem + sizeof(ha::smem) / sizeof(float), nanv);
  }
  BlockCtx ctx;
  std::barrier<> block_bar(nthreads);
  ctx.block_bar = &block_bar;
  for (unsigned w = 0; w < nwaves; ++w) {
    ctx.wave_bar.emplace_back(new std::barrier<>(64));
    ctx.xch.emplace_back(128, 0);
  }
  const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
  std::vector<std::thread> threads;
  threads.reserve(nthreads);
  for (unsigned t = 0; t < nthreads; ++t) {
    threads.emplace_back([&, t]() {
      t_ctx = &ctx;
      t_blockDim = block;
      t_gridDim = grid;
      t_threadIdx = dim3(t, 0, 0);
      for (unsigned long long b = 0; b < nblocks; ++b) {
        t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y)));
        body();
        // a thread that returned early from the kernel must not leave the others stuck: kernels under test
        // only return early wave-uniformly and never before a later barrier, so a plain block barrier here
        // keeps blocks from overlapping in the shared LDS array.
        ctx.block_bar->arrive_and_wait();
      }
    });
  }
  for (auto& th : threads) th.join();
}
}  // namespace simt_emu
