// does ANY load flavour make the L2 issue fabric reads smaller than 128 B for the 8-byte entries of a hashed level?
// Stand-alone model of K2's gathers (no MLP): 2^19-entry levels of 8-byte entries (F = 4 halfs), 8 trilinear corners per sample and level through the
// instant-ngp hash, positions uniform in the unit cube (the fine levels of a trained scene are no more coherent than this: the hash spreads them), 8 independent
// gathers in flight per lane -- one template instance per flavour so that rocprofv3 --pmc TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum reports them separately.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/k2_request_size tools/k2_request_size.hip        run: tools/k2_request_size [n_samples] [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t N_LEVELS = 5, LOG2_T = 19, T = 1u << LOG2_T;
enum Flavour { PLAIN = 0, SC1, NT, SC0_SC1, SC0_SC1_NT, BUFFER, BUFFER_SC1, SCALAR, N_FLAVOURS };
static const char* kNames[] = {"global_load_dwordx2", "global_load_dwordx2 sc1", "global_load_dwordx2 nt", "global_load_dwordx2 sc0 sc1", "global_load_dwordx2 sc0 sc1 nt",
	"buffer_load_dwordx2 offen", "buffer_load_dwordx2 offen sc1", "s_load_dwordx2 (one lane's address at a time)"};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// eight gathers in flight and the wait for them in ONE asm statement (separate statements would let the compiler reuse an output register before its data has arrived)
#define GL8(MOD) asm volatile( \
	"global_load_dwordx2 %0, %8, off " MOD "\n\tglobal_load_dwordx2 %1, %9, off " MOD "\n\tglobal_load_dwordx2 %2, %10, off " MOD "\n\tglobal_load_dwordx2 %3, %11, off " MOD "\n\t" \
	"global_load_dwordx2 %4, %12, off " MOD "\n\tglobal_load_dwordx2 %5, %13, off " MOD "\n\tglobal_load_dwordx2 %6, %14, off " MOD "\n\tglobal_load_dwordx2 %7, %15, off " MOD "\n\ts_waitcnt vmcnt(0)" \
	: "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
	: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory")
#define BL8(MOD) asm volatile( \
	"buffer_load_dwordx2 %0, %8, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %1, %9, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %2, %10, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %3, %11, %16, 0 offen " MOD "\n\t" \
	"buffer_load_dwordx2 %4, %12, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %5, %13, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %6, %14, %16, 0 offen " MOD "\n\tbuffer_load_dwordx2 %7, %15, %16, 0 offen " MOD "\n\ts_waitcnt vmcnt(0)" \
	: "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
	: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "s"(rsrc) : "memory")
template <int FL> __device__ __forceinline__ void load8(const uint2* base, const uint32_t (&idx)[8], u32x4 rsrc, u32x2 (&v)[8]) {
	const uint2* p[8]; uint32_t o[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) { p[c] = base + idx[c]; o[c] = idx[c] * 8u; }
	if constexpr (FL == PLAIN) GL8("");
	else if constexpr (FL == SC1) GL8("sc1");
	else if constexpr (FL == NT) GL8("nt");
	else if constexpr (FL == SC0_SC1) GL8("sc0 sc1");
	else if constexpr (FL == SC0_SC1_NT) GL8("sc0 sc1 nt");
	else if constexpr (FL == BUFFER) BL8("");
	else if constexpr (FL == BUFFER_SC1) BL8("sc1");
}

template <int FL, int ALLOC /* 0 hipMalloc, 1 uncached, 2 fine-grained: a name of its own in the PMC pass */>
__global__ void __launch_bounds__(256) k_gather(const uint2* __restrict__ table, const float* __restrict__ pos, uint32_t n, uint2* __restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float px = pos[i * 3 + 0], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
	uint32_t ax = 0, ay = 0;
	for (uint32_t l = 0; l < N_LEVELS; ++l) {
		const uint2* lv = table + (size_t)l * T;
		const float scale = (float)(128u << l) - 1.0f; // levels 3..7 of base.json (N_min 16, b = 2)
		const uint32_t x = (uint32_t)(px * scale + 0.5f), y = (uint32_t)(py * scale + 0.5f), z = (uint32_t)(pz * scale + 0.5f);
		uint32_t idx[8];
#pragma unroll
		for (uint32_t c = 0; c < 8; ++c) idx[c] = ((x + (c & 1u)) ^ ((y + ((c >> 1) & 1u)) * 2654435761u) ^ ((z + (c >> 2)) * 805459861u)) & (T - 1u);
		uint2 v[8];
		if constexpr (FL == SCALAR) {
			// the scalar cache has 64-byte lines: one lane's eight addresses at a time through s_load (a request-size probe, not a candidate)
#pragma unroll
			for (uint32_t c = 0; c < 8; ++c) v[c] = make_uint2(0u, 0u);
			for (uint32_t ln = 0; ln < 64; ++ln) {
#pragma unroll
				for (uint32_t c = 0; c < 8; ++c) {
					const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)idx[c], (int)ln) & (T - 1u); // (a lane past n has left: its registers are not indices)
					const uint2* p = lv + id; u32x2 s;
					asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(p) : "memory");
					if ((threadIdx.x & 63u) == ln) v[c] = make_uint2(s.x, s.y);
				}
			}
		} else {
			u32x4 rsrc; const uint64_t a = (uint64_t)lv;
			rsrc.x = (uint32_t)a; rsrc.y = (uint32_t)(a >> 32) & 0xffffu; rsrc.z = T * 8u; rsrc.w = 0x00020000u; // raw buffer, DATA_FORMAT = 32 (gfx9 encoding)
			rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y); rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
			u32x2 w[8]; load8<FL>(lv, idx, rsrc, w);
#pragma unroll
			for (uint32_t c = 0; c < 8; ++c) v[c] = make_uint2(w[c].x, w[c].y);
		}
#pragma unroll
		for (uint32_t c = 0; c < 8; ++c) { ax ^= v[c].x; ay += v[c].y; }
	}
	out[i] = make_uint2(ax, ay);
}

template <int FL, int ALLOC = 0> static float run(const uint2* table, const float* pos, uint32_t n, uint2* out, int reps, hipStream_t s) {
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const uint32_t grid = (n + 255u) / 256u;
	for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_gather<FL, ALLOC>), dim3(grid), dim3(256), 0, s, table, pos, n, out);
	CHK(hipEventRecord(e0, s));
	for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_gather<FL, ALLOC>), dim3(grid), dim3(256), 0, s, table, pos, n, out);
	CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps * 1000.f;
}

int main(int argc, char** argv) {
	const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 380000u; const int reps = argc > 2 ? atoi(argv[2]) : 20;
	hipStream_t s; CHK(hipStreamCreate(&s));
	std::vector<float> pos((size_t)n * 3); uint64_t st = 0x853c49e6748fea9bull;
	for (auto& p : pos) { st = st * 6364136223846793005ull + 1442695040888963407ull; p = (float)((st >> 40) & 0xffffff) / 16777216.0f; }
	std::vector<uint32_t> tab((size_t)N_LEVELS * T * 2); for (size_t k = 0; k < tab.size(); ++k) tab[k] = (uint32_t)(k * 2654435761u);
	float* d_pos; uint2 *d_out, *d_tab[3]; const char* alloc_names[3] = {"hipMalloc", "hipExtMallocWithFlags(hipDeviceMallocUncached)", "hipExtMallocWithFlags(hipDeviceMallocFinegrained)"};
	CHK(hipMalloc(&d_pos, pos.size() * 4)); CHK(hipMalloc(&d_out, (size_t)n * 8)); CHK(hipMemcpy(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
	CHK(hipMalloc(&d_tab[0], tab.size() * 4));
	if (hipExtMallocWithFlags((void**)&d_tab[1], tab.size() * 4, hipDeviceMallocUncached) != hipSuccess) d_tab[1] = nullptr;
	if (hipExtMallocWithFlags((void**)&d_tab[2], tab.size() * 4, hipDeviceMallocFinegrained) != hipSuccess) d_tab[2] = nullptr;
	for (int a = 0; a < 3; ++a) if (d_tab[a]) CHK(hipMemcpy(d_tab[a], tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
	const double alg_mb = (double)n * N_LEVELS * 8 * 8 / 1e6;
	setvbuf(stdout, nullptr, _IOLBF, 0);
	printf("# %u samples x %u levels x 8 corners x 8 B = %.1f MB of entries per launch; launch order below = dispatch order in the PMC pass (3 warm-up + %d timed launches each)\n", n, N_LEVELS, alg_mb, reps);
	for (int a = 0; a < 3; ++a) {
		if (!d_tab[a]) { printf("%-52s allocation refused\n", alloc_names[a]); continue; }
		float us[N_FLAVOURS];
		us[PLAIN] = a == 0 ? run<PLAIN, 0>(d_tab[a], d_pos, n, d_out, reps, s) : a == 1 ? run<PLAIN, 1>(d_tab[a], d_pos, n, d_out, reps, s) : run<PLAIN, 2>(d_tab[a], d_pos, n, d_out, reps, s);
		if (a == 0) {
			us[SC1] = run<SC1>(d_tab[a], d_pos, n, d_out, reps, s); us[NT] = run<NT>(d_tab[a], d_pos, n, d_out, reps, s);
			us[SC0_SC1] = run<SC0_SC1>(d_tab[a], d_pos, n, d_out, reps, s); us[SC0_SC1_NT] = run<SC0_SC1_NT>(d_tab[a], d_pos, n, d_out, reps, s);
			us[BUFFER] = run<BUFFER>(d_tab[a], d_pos, n, d_out, reps, s); us[BUFFER_SC1] = run<BUFFER_SC1>(d_tab[a], d_pos, n, d_out, reps, s);
			us[SCALAR] = run<SCALAR>(d_tab[a], d_pos, n, d_out, std::max(1, reps / 10), s);
		}
		for (int f = 0; f < (a == 0 ? (int)N_FLAVOURS : 1); ++f) printf("%-52s %-46s %8.1f us  %7.1f GB/s of entries\n", alloc_names[a], kNames[f], us[f], alg_mb / us[f] * 1e3);
	}
	return 0;
}
