// oracle/ref_shim: the pcg32 generator (M. O'Neill's PCG-XSH-RR 64/32, as wrapped by W. Jakob's pcg32.h which the reference includes; absent from the mount): the member
// functions the reference's headers call, restated from the published algorithm -- the same restatement oracle/ora_math.hpp pins against the published known-answer values.
#pragma once
#include <cstdint>
#define PCG32_DEFAULT_STATE 0x853c49e6748fea9bULL
#define PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PCG32_MULT 0x5851f42d4c957f2dULL
struct pcg32 {
	uint64_t state = PCG32_DEFAULT_STATE, inc = PCG32_DEFAULT_STREAM;
	pcg32() {}
	pcg32(uint64_t initstate, uint64_t initseq = 1u) { seed(initstate, initseq); }
	void seed(uint64_t initstate, uint64_t initseq = 1) { state = 0u; inc = (initseq << 1u) | 1u; next_uint(); state += initstate; next_uint(); }
	uint32_t next_uint() {
		const uint64_t oldstate = state;
		state = oldstate * PCG32_MULT + inc;
		const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u), rot = (uint32_t)(oldstate >> 59u);
		return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
	}
	uint32_t next_uint(uint32_t bound) { const uint32_t threshold = (~bound + 1u) % bound; for (;;) { const uint32_t r = next_uint(); if (r >= threshold) return r % bound; } }
	float next_float() { union { uint32_t u; float f; } x; x.u = (next_uint() >> 9) | 0x3f800000u; return x.f - 1.0f; }
	void advance(int64_t delta_ = (1ll << 32)) {
		uint64_t cur_mult = PCG32_MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u, delta = (uint64_t)delta_;
		while (delta > 0) { if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; } cur_plus = (cur_mult + 1) * cur_plus; cur_mult *= cur_mult; delta /= 2; }
		state = acc_mult * state + acc_plus;
	}
};
