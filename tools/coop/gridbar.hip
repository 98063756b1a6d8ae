// What does a grid-wide barrier cost on MI355X, and is a cooperative launch available on this pool?
//   hipcc --offload-arch=gfx950 -O2 -o gridbar gridbar.hip && ./gridbar
// Per variant: a persistent kernel of G workgroups x 512 threads runs K phases; in every phase a workgroup writes one
// 64x32 f32 tile (nontemporal stores, as the product's plane stores), meets the others at a grid barrier and reads the tile
// of workgroup (b + 97) % G, which lives on another XCD: the sum it reads proves visibility.  Printed: us per phase.
// Variants: cg (cooperative_groups grid.sync()), flat (one counter + generation word), xcd (a counter per XCD, the last
// arrival of an XCD counts up the device counter), and the same work as K plain launches on one stream for comparison.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NT = 512, TILE = 64 * 32;

struct Bar {
    unsigned* cnt;   // [0]: device counter, [16 * (1 + x)]: XCD x
    unsigned* gen;   // generation word
};

__device__ __forceinline__ void bar_flat(Bar b, unsigned G, unsigned& g) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned want = g + 1;
        __atomic_thread_fence(__ATOMIC_RELEASE);  // (system scope by default in HIP: agent is enough, see bar_flat_agent)
        const unsigned old = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            __hip_atomic_store(b.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(b.gen, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    g = g + 1;
    __syncthreads();
}
__device__ __forceinline__ void bar_agent(Bar b, unsigned G, unsigned& g) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned want = g + 1;
        const unsigned old = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            __hip_atomic_store(b.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(b.gen, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    g = g + 1;
    __syncthreads();
}
__device__ __forceinline__ void bar_xcd(Bar b, unsigned G, unsigned& g) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned want = g + 1;
        const unsigned x = blockIdx.x & 7u, per = (G >> 3) + ((G & 7u) > x ? 1u : 0u);
        const unsigned old = __hip_atomic_fetch_add(b.cnt + 16 * (1 + x), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        bool last = false;
        if (old == per - 1) {
            __hip_atomic_store(b.cnt + 16 * (1 + x), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned o2 = __hip_atomic_fetch_add(b.cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = o2 == 7;
        }
        if (last) {
            __hip_atomic_store(b.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(b.gen, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    g = g + 1;
    __syncthreads();
}

// MODE 0 cg, 1 flat (system fences), 2 flat agent, 3 per-XCD; WORK: tile write + remote tile read per phase
template <int MODE, bool WORK>
__global__ void __launch_bounds__(NT) k_phases(float* planeA, float* planeB, Bar b, int K, unsigned* bad, unsigned gen0) {
    const unsigned G = gridDim.x, bid = blockIdx.x, tid = threadIdx.x;
    unsigned g = gen0;
    float* src = planeA;
    float* dst = planeB;
    unsigned errs = 0;
    for (int k = 0; k < K; ++k) {
        if (WORK) {
            float4 v = make_float4((float)(k + 1), (float)bid, 1.0f, (float)tid);
            __builtin_nontemporal_store(v.x, dst + (size_t)bid * TILE + 4 * tid + 0);
            __builtin_nontemporal_store(v.y, dst + (size_t)bid * TILE + 4 * tid + 1);
            __builtin_nontemporal_store(v.z, dst + (size_t)bid * TILE + 4 * tid + 2);
            __builtin_nontemporal_store(v.w, dst + (size_t)bid * TILE + 4 * tid + 3);
        }
        if (MODE == 0) cg::this_grid().sync();
        else if (MODE == 1) bar_flat(b, G, g);
        else if (MODE == 2) bar_agent(b, G, g);
        else bar_xcd(b, G, g);
        if (WORK) {
            const unsigned o = (bid + 97u) % G;
            const float4 r = *reinterpret_cast<const float4*>(dst + (size_t)o * TILE + 4 * tid);
            if (r.x != (float)(k + 1) || r.y != (float)o || r.w != (float)tid) ++errs;
        }
        float* t = src; src = dst; dst = t;
    }
    if (errs) atomicAdd(bad, errs);
}
__global__ void __launch_bounds__(NT) k_one(float* dst, int k) {
    const unsigned bid = blockIdx.x, tid = threadIdx.x;
    float4 v = make_float4((float)(k + 1), (float)bid, 1.0f, (float)tid);
    *reinterpret_cast<float4*>(dst + (size_t)bid * TILE + 4 * tid) = v;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 200;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, cooperativeLaunch=%d\n", prop.name, prop.multiProcessorCount, prop.cooperativeLaunch);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int maxG = 2048;
    float *A, *B;
    CK(hipMalloc(&A, (size_t)maxG * TILE * 4));
    CK(hipMalloc(&B, (size_t)maxG * TILE * 4));
    unsigned *cnt, *gen, *bad;
    CK(hipMalloc(&cnt, 4096));
    CK(hipMalloc(&gen, 256));
    CK(hipMalloc(&bad, 256));
    CK(hipMemset(cnt, 0, 4096));
    CK(hipMemset(gen, 0, 256));
    CK(hipMemset(bad, 0, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int occ[4] = {0, 0, 0, 0};
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[0], k_phases<0, true>, NT, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[2], k_phases<2, true>, NT, 0));
    printf("occupancy (workgroups of %d per CU): cg %d, flat %d\n", NT, occ[0], occ[2]);
    unsigned gen_host = 0;
    auto run = [&](int mode, bool work, int G, bool coop) -> int {
        Bar b{cnt, gen};
        int k = K;
        unsigned g0 = gen_host;
        void* args[] = {&A, &B, &b, &k, &bad, &g0};
        const void* fn = nullptr;
#define PICK(M) (work ? (const void*)k_phases<M, true> : (const void*)k_phases<M, false>)
        fn = mode == 0 ? PICK(0) : mode == 1 ? PICK(1) : mode == 2 ? PICK(2) : PICK(3);
        for (int rep = 0; rep < 3; ++rep) {
            g0 = gen_host;
            CK(hipEventRecord(e0, s));
            hipError_t e = coop ? hipLaunchCooperativeKernel(fn, dim3(G), dim3(NT), args, 0, s)
                                : hipLaunchKernel(fn, dim3(G), dim3(NT), args, 0, s);
            if (e != hipSuccess) { printf("  launch failed: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return 0; }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            if (mode != 0) gen_host += K;
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hb = 0;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep == 2)
                printf("  mode %d %-5s %-4s G=%4d: %7.2f us per phase (%d phases, %.3f ms)  visibility errors %u\n", mode,
                       coop ? "coop" : "plain", work ? "work" : "bare", G, ms * 1e3f / K, K, ms, hb);
        }
        return 0;
    };
    const int Gs[] = {64, 256, 512, 768};
    for (int G : Gs) {
        if (G > prop.multiProcessorCount * occ[0]) continue;
        for (int work = 0; work < 2; ++work) {
            run(0, work, G, true);
            run(1, work, G, true);
            run(2, work, G, true);
            run(3, work, G, true);
            run(2, work, G, false);  // the same spin barrier under a plain launch (co-residency by grid size only)
            run(3, work, G, false);
        }
    }
    // one cooperative launch with one phase: what the launch itself costs against a plain one
    for (int coop = 0; coop < 2; ++coop) {
        Bar b{cnt, gen};
        int k = 1;
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            unsigned g0 = gen_host;
            void* args[] = {&A, &B, &b, &k, &bad, &g0};
            CK(hipEventRecord(e0, s));
            for (int j = 0; j < 10; ++j) {
                g0 = gen_host;
                hipError_t e = coop ? hipLaunchCooperativeKernel((const void*)k_phases<2, true>, dim3(256), dim3(NT), args, 0, s)
                                    : hipLaunchKernel((const void*)k_phases<2, true>, dim3(256), dim3(NT), args, 0, s);
                if (e != hipSuccess) { printf("launch: %s\n", hipGetErrorString(e)); return 1; }
                gen_host += 1;
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("10 back-to-back %s launches of a 1-phase kernel (G=256): %.2f us each\n", coop ? "cooperative" : "plain", best * 100.0f);
    }
    // the same tile writes as K dependent plain launches
    for (int G : Gs) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_one, dim3(G), dim3(NT), 0, s, (k & 1) ? A : B, k);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("  %d dependent plain launches, G=%4d: %7.2f us per launch\n", K, G, best * 1e3f / K);
    }
    return 0;
}
