// Round 5: what does a gather cost K2 -- its line, or its lane address?  tools/k2_request_size.hip showed 15.2 M random 8-byte gathers (5 hashed levels) in 91 us moving 690 MB of
// 128-byte lines (7.6 TB/s); K2 proper issues 24 M gathers (8 levels, 3 of them dense tables that live in L1 / L2) in 120 us for 406 MB of lines.  Both are ~0.7 - 0.8 lane
// addresses per ns and CU.  If the address rate is the bound, the dense levels' gathers (37 % of the addresses, none of the traffic) are worth removing:
//   v0  5 hashed levels, 8 x 8-byte gathers each                       (= k2_request_size's plain flavour)
//   v1  v0 + the 3 dense levels (16^3, 32^3, 64^3), 8 x 8-byte gathers each   (= K2's address stream)
//   v2  v0 + the dense levels with the x-adjacent corner pair in ONE 16-byte load (4 addresses per level; 8-byte aligned)
//   v3  v0 + level 16^3 (32 KiB) from LDS + the other two dense levels as in v2
//   v4  the 3 dense levels alone, 8 x 8-byte gathers
//   v5  the 3 dense levels alone, pair loads
// build: hipcc --offload-arch=gfx950 -O3 -o tools/k2_lane_address tools/k2_lane_address.hip      run: tools/k2_lane_address [n_samples] [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr uint32_t T = 1u << 19, N_HASHED = 5;
constexpr uint32_t DENSE_RES[3] = {17u, 33u, 65u}; // grid points per axis of levels 0..2 (resolution + 1)
constexpr uint32_t DENSE_OFF[3] = {0u, 17u * 17u * 17u, 17u * 17u * 17u + 33u * 33u * 33u};
constexpr uint32_t DENSE_TOTAL = 17u * 17u * 17u + 33u * 33u * 33u + 65u * 65u * 65u;

template <int V>
__global__ void __launch_bounds__(256) k_gather(const uint2* __restrict__ hashed, const uint2* __restrict__ dense, const float* __restrict__ pos, uint32_t n, uint2* __restrict__ out) {
	__shared__ uint2 s_l0[17 * 17 * 17];
	if (V == 3) { for (uint32_t k = threadIdx.x; k < 17u * 17u * 17u; k += 256u) s_l0[k] = dense[k]; __syncthreads(); }
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float px = pos[i * 3 + 0], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
	uint32_t ax = 0, ay = 0;
	if (V <= 3) for (uint32_t l = 0; l < N_HASHED; ++l) {
		const uint2* lv = hashed + (size_t)l * T;
		const float scale = (float)(128u << l) - 1.0f;
		const uint32_t x = (uint32_t)(px * scale + 0.5f), y = (uint32_t)(py * scale + 0.5f), z = (uint32_t)(pz * scale + 0.5f);
		uint2 v[8];
#pragma unroll
		for (uint32_t c = 0; c < 8; ++c) v[c] = lv[((x + (c & 1u)) ^ ((y + ((c >> 1) & 1u)) * 2654435761u) ^ ((z + (c >> 2)) * 805459861u)) & (T - 1u)];
#pragma unroll
		for (uint32_t c = 0; c < 8; ++c) { ax ^= v[c].x; ay += v[c].y; }
	}
	if (V >= 1) {
#pragma unroll
		for (uint32_t l = 0; l < 3; ++l) {
			const uint32_t R = DENSE_RES[l];
			const uint2* lv = dense + DENSE_OFF[l];
			const float scale = (float)(R - 1u) - 1e-3f;
			const uint32_t x = (uint32_t)(px * scale), y = (uint32_t)(py * scale), z = (uint32_t)(pz * scale);
			const uint32_t base = x + R * (y + R * z);
			if (V == 1 || V == 4) {
				uint2 v[8];
#pragma unroll
				for (uint32_t c = 0; c < 8; ++c) v[c] = lv[base + (c & 1u) + R * (((c >> 1) & 1u) + R * (c >> 2))];
#pragma unroll
				for (uint32_t c = 0; c < 8; ++c) { ax ^= v[c].x; ay += v[c].y; }
			} else if (V == 3 && l == 0) {
				uint2 v[8];
#pragma unroll
				for (uint32_t c = 0; c < 8; ++c) v[c] = s_l0[base + (c & 1u) + R * (((c >> 1) & 1u) + R * (c >> 2))];
#pragma unroll
				for (uint32_t c = 0; c < 8; ++c) { ax ^= v[c].x; ay += v[c].y; }
			} else {
				typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
				u32x4 v[4]; const uint2* p[4];
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) p[c] = lv + base + R * ((c & 1u) + R * (c >> 1));
				// four loads in flight and their wait in ONE statement (an output register must not be reused before its data has arrived)
				asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
					: "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) { ax ^= v[c].x ^ v[c].z; ay += v[c].y + v[c].w; }
			}
		}
	}
	out[i] = make_uint2(ax, ay);
}
template <int V> static float run(const uint2* hashed, const uint2* dense, const float* pos, uint32_t n, uint2* out, int reps, hipStream_t s) {
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const uint32_t grid = (n + 255u) / 256u;
	for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_gather<V>, dim3(grid), dim3(256), 0, s, hashed, dense, pos, n, out);
	CHK(hipEventRecord(e0, s));
	for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_gather<V>, dim3(grid), dim3(256), 0, s, hashed, dense, pos, n, out);
	CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps * 1000.f;
}
int main(int argc, char** argv) {
	setvbuf(stdout, nullptr, _IOLBF, 0);
	const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 380000u; const int reps = argc > 2 ? atoi(argv[2]) : 20;
	hipStream_t s; CHK(hipStreamCreate(&s));
	std::vector<float> pos((size_t)n * 3); uint64_t st = 0x853c49e6748fea9bull;
	for (auto& p : pos) { st = st * 6364136223846793005ull + 1442695040888963407ull; p = (float)((st >> 40) & 0xffffff) / 16777216.0f; }
	std::vector<uint32_t> tab((size_t)N_HASHED * T * 2 + (size_t)(DENSE_TOTAL + 8) * 2); for (size_t k = 0; k < tab.size(); ++k) tab[k] = (uint32_t)(k * 2654435761u);
	float* d_pos; uint2 *d_out, *d_tab;
	CHK(hipMalloc(&d_pos, pos.size() * 4)); CHK(hipMalloc(&d_out, (size_t)n * 8)); CHK(hipMalloc(&d_tab, tab.size() * 4));
	CHK(hipMemcpy(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
	const uint2* hashed = d_tab; const uint2* dense = d_tab + (size_t)N_HASHED * T;
	const char* names[6] = {"v0 5 hashed levels (40 addresses / sample)", "v1 + 3 dense levels, 8 x 8 B each (64 addresses)", "v2 + 3 dense levels, 4 x 16 B pair loads (52 addresses)",
		"v3 + level 0 from LDS, levels 1-2 pair loads (48 addresses + 8 LDS reads)", "v4 3 dense levels alone, 8 x 8 B (24 addresses)", "v5 3 dense levels alone, pair loads (12 addresses)"};
	float us[6];
	for (int rep = 0; rep < 2; ++rep) {
		us[0] = run<0>(hashed, dense, d_pos, n, d_out, reps, s); us[1] = run<1>(hashed, dense, d_pos, n, d_out, reps, s); us[2] = run<2>(hashed, dense, d_pos, n, d_out, reps, s);
		us[3] = run<3>(hashed, dense, d_pos, n, d_out, reps, s); us[4] = run<4>(hashed, dense, d_pos, n, d_out, reps, s); us[5] = run<5>(hashed, dense, d_pos, n, d_out, reps, s);
		for (int v = 0; v < 6; ++v) printf("pass %d  %-78s %8.1f us\n", rep, names[v], us[v]);
	}
	return 0;
}
