// 2-bit packed sequence helpers shared by the device kernels.
// Layout: 32 bases per 64-bit word, base i of a stream at bits [2*(i&31), 2*(i&31)+2) of word i>>5
// (A=0 C=1 G=2 T=3).  "Spaced" masks use the same geometry with one flag bit at the even bit
// of each base.  Database arrays carry one zero word in front of element 0, so a window that
// starts up to 32 bases left of the database is addressable.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgx {

constexpr uint64_t kEven = 0x5555555555555555ull;
constexpr int kSeedK = 16;                    // probe k-mer: 16 bases = one 32-bit key
constexpr int kWord = 28;                     // megablast word size W (spec S3)
constexpr int kProbeStride = kWord - kSeedK + 1; // 13: every exact 28-mer holds one probe
constexpr int kXdrop = 10;
constexpr int kBlkShift = 9;                   // blk_subj[b] = subject holding base b << 9 (512-base blocks)

// 64 bits (32 bases) of a packed stream starting at base position `pos` (pos >= -32)
__device__ __forceinline__ uint64_t window64(const uint64_t *words, int64_t pos)
{
	int64_t wi = pos >> 5; // arithmetic shift: floor
	int sh = (int)(pos & 31) * 2;
	uint64_t lo = words[wi];
	if (sh == 0)
		return lo;
	uint64_t hi = words[wi + 1];
	return (lo >> sh) | (hi << (64 - sh));
}

__device__ __forceinline__ uint32_t kmer16(const uint64_t *words, int64_t pos)
{
	return (uint32_t)window64(words, pos);
}

// 16 bases starting at base `pos` (pos >= -32): ONE 8-byte load at 4-byte alignment plus one funnel shift
// (64 - 2*(pos & 15) >= 34 valid bits, so the low 32 are always whole)
struct __attribute__((packed, aligned(4))) U64Align4 {
	uint64_t v;
};
__device__ __forceinline__ uint32_t window16(const uint64_t *words, int64_t pos)
{
	const uint32_t *d = reinterpret_cast<const uint32_t *>(words);
	const uint64_t v = reinterpret_cast<const U64Align4 *>(d + (pos >> 4))->v;
	return (uint32_t)(v >> ((int)(pos & 15) * 2));
}

__host__ __device__ __forceinline__ uint32_t seed_bucket(uint32_t kmer, int bits)
{
	return bits >= 32 ? kmer : (uint32_t)((kmer * 0x9E3779B1u) >> (32 - bits));
}

// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ uint64_t reverse_groups(uint64_t x)
{
	uint64_t y = __brevll(x);
	return ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & kEven) << 1);
}

// spaced mask with flags for bases [a, b) of a word (0 <= a <= b <= 32)
__device__ __forceinline__ uint64_t spaced_range(int a, int b)
{
	if (b <= a)
		return 0;
	uint64_t hi = (b >= 32) ? ~0ull : ((1ull << (2 * b)) - 1);
	uint64_t lo = (a <= 0) ? 0ull : ((a >= 32) ? ~0ull : ((1ull << (2 * a)) - 1));
	return (hi & ~lo) & kEven;
}

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint64_t synth_hash(uint64_t seed, uint64_t tag, uint64_t i, uint64_t j)
{
	return splitmix64(splitmix64(splitmix64(seed + tag) + i) + j);
}

} // namespace pgx
