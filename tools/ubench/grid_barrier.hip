// grid_barrier.hip -- development microbenchmark: what does one grid barrier of a cooperative launch cost on
// an MI355X, as a function of the number of resident blocks and of how the barrier is built?  The cooperative cells spend three
// phases of ~14 us each at quarter resolution for 18 MB of traffic; two of the phase boundaries are grid barriers (a phase trace,
// tools/trace_coop.py, puts 8.0 + 8.9 of the launch's 37 us into them).
//
//   flat        one arrival counter + one generation word, lane-0 release fence before the arrival, acquire fence after the release
//               (urnn_common.h coop_grid_barrier as of round 5)
//   shard8      arrival counters sharded 8 ways (blockIdx % 8), the last arriver of a shard bumps a top counter, the last of those
//               writes eight generation words (256 B apart), a block polls the word of its shard; fences as flat
//   flat_nf     flat without the two fences: what is published across the barrier travels as 8-byte agent-scope atomic stores and is
//               read back with agent-scope atomic loads (MI355X_MICROARCH.md "valid forms": 8-B agent atomics on both sides)
//   shard8_nf   shard8 without fences
//   shard16_nf  16 shards of <= 16 arrivals, 16 generation words
//   shard8_nf1  shard8_nf with ONE generation word for all pollers
//
// Each kernel runs R barriers back to back; in the *_nf variants every block publishes an {epoch, block} granule before each barrier
// and wave 0 reads ALL blocks' granules after it and counts stale ones (must be 0).  The host divides the kernel's duration minus that
// of an R = 0 launch by R.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

enum { FLAT = 0, SHARD8 = 1, FLAT_NF = 2, SHARD8_NF = 3, SHARD16_NF = 4, SHARD8_NF1 = 5, NVAR = 6 };
static const char *names[NVAR] = {"flat", "shard8", "flat_nf", "shard8_nf", "shard16_nf", "shard8_nf1"};

// bar layout (dwords, 64 apart = 256 B): [0] arrivals, [64] generation, [128 + 64 s] shard arrivals (s < 16), [1152] top counter,
// [1216 + 64 s] shard generation words
template <int VAR>
__device__ __forceinline__ void barrier(unsigned *bar, unsigned nblocks)
{
    constexpr bool FENCE = VAR == FLAT || VAR == SHARD8;
    constexpr int NS = VAR == SHARD16_NF ? 16 : 8;
    constexpr bool SHARDED = VAR != FLAT && VAR != FLAT_NF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = blockIdx.x % NS;
        unsigned *genw = (SHARDED && VAR != SHARD8_NF1) ? bar + 1216 + 64 * s : bar + 64;
        const unsigned gen = __hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        bool last;
        if (!SHARDED) {
            const unsigned t = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == nblocks - 1;
            if (last) __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned mine = nblocks / NS + (s < nblocks % NS ? 1u : 0u);      // blocks with blockIdx % NS == s
            const unsigned used = nblocks < NS ? nblocks : (unsigned)NS;
            const unsigned t = __hip_atomic_fetch_add(&bar[128 + 64 * s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = false;
            if (t == mine - 1) {
                __hip_atomic_store(&bar[128 + 64 * s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned u = __hip_atomic_fetch_add(&bar[1152], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (u == used - 1) {
                    __hip_atomic_store(&bar[1152], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = true;
                }
            }
        }
        if (last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the counter resets have left before anybody is released
            if (SHARDED && VAR != SHARD8_NF1) {
                for (int k = 0; k < NS; ++k) __hip_atomic_store(bar + 1216 + 64 * k, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(&bar[64], gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            while (__hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int VAR>
__global__ __launch_bounds__(768) void bench_kernel(unsigned *bar, unsigned long long *slots, int rounds, unsigned *errors)
{
    constexpr bool FENCE = VAR == FLAT || VAR == SHARD8;
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) {
            const unsigned long long g = ((unsigned long long)(r + 1) << 32) | blockIdx.x;
            if (FENCE) slots[blockIdx.x] = g;                                                    // plain store, published by the release fence
            else __hip_atomic_store(&slots[blockIdx.x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        barrier<VAR>(bar, gridDim.x);
        if (threadIdx.x < 64) {
            for (unsigned b = threadIdx.x; b < gridDim.x; b += 64) {
                const unsigned long long g = FENCE ? slots[b] : __hip_atomic_load(&slots[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((g >> 32) != (unsigned)(r + 1) || (unsigned)g != b) ++bad;
            }
        }
        // the read side is done before anybody publishes round r + 1: the next barrier's arrival needs every block here first ... but a
        // fast block may overwrite its slot while a slow one still reads -> a second barrier, as the real kernels have (two per phase pair)
        barrier<VAR>(bar, gridDim.x);
    }
    if (bad) atomicAdd(errors, bad);
}

template <int VAR>
static float run(unsigned *bar, unsigned long long *slots, unsigned *errors, int blocks, int threads, int rounds, int reps)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int i = 0; i < reps + 2; ++i) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(bench_kernel<VAR>, dim3(blocks), dim3(threads), 0, 0, bar, slots, rounds, errors);
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 2 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    unsigned *bar, *errors;
    unsigned long long *slots;
    (void)hipMalloc(&bar, 4096 * 4);
    (void)hipMemset(bar, 0, 4096 * 4);
    (void)hipMalloc(&slots, 1024 * 8);
    (void)hipMemset(slots, 0, 1024 * 8);
    (void)hipMalloc(&errors, 64);
    (void)hipMemset(errors, 0, 64);
    const int R = 100;      // 2 R barriers per launch
    printf("%-11s %7s %8s %12s %12s %8s\n", "variant", "blocks", "threads", "empty_us", "us/barrier", "stale");
    for (int threads : {768, 256}) {
        for (int blocks : {32, 64, 128, 245, 256}) {
            float e[NVAR], t[NVAR];
            unsigned err[NVAR];
#define RUN(V) do { (void)hipMemset(errors, 0, 4); e[V] = run<V>(bar, slots, errors, blocks, threads, 0, 5); t[V] = run<V>(bar, slots, errors, blocks, threads, R, 5); \
                    (void)hipMemcpy(&err[V], errors, 4, hipMemcpyDeviceToHost); } while (0)
            RUN(FLAT); RUN(SHARD8); RUN(FLAT_NF); RUN(SHARD8_NF); RUN(SHARD16_NF); RUN(SHARD8_NF1);
            for (int v = 0; v < NVAR; ++v) printf("%-11s %7d %8d %12.2f %12.3f %8u\n", names[v], blocks, threads, e[v], (t[v] - e[v]) / (2 * R), err[v]);
        }
    }
    unsigned h[4096];
    (void)hipMemcpy(h, bar, sizeof(h), hipMemcpyDeviceToHost);
    printf("counters after the runs (must be 0): %u %u %u\n", h[0], h[128], h[1152]);
    return 0;
}
