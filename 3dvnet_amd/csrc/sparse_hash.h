// Open-addressing hash table over packed (batch, x, y, z) sparse-tensor coordinates: device-side lookup shared by
// sparse.hip (table build, neighbour tables, interpolation) and decoder.hip (fused hypothesis decoder).
#pragma once
#include "v3d_common.h"

namespace v3dhash {

constexpr unsigned long long kEmpty = ~0ull;
constexpr int kGuard = 8;   // coordinates may be probed a few voxels below zero

__device__ __forceinline__ unsigned long long pack_key(int b, int x, int y, int z) {
  return ((unsigned long long)(unsigned)(b & 0xffff) << 48) | ((unsigned long long)(unsigned)((x + kGuard) & 0xffff) << 32) |
         ((unsigned long long)(unsigned)((y + kGuard) & 0xffff) << 16) | (unsigned long long)(unsigned)((z + kGuard) & 0xffff);
}

__device__ __forceinline__ unsigned hash_u64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned)k;
}

// One 16-byte entry per slot: a lookup is ONE memory request whether the key is present or not.  (Keys and values in two arrays --
// rounds 1-4 -- cost a second, dependent request per hit; the lookup kernels are bound by the number of L2 requests)
struct __attribute__((aligned(16))) HashEntry {
  unsigned long long key;
  int val;
  int pad;
};

struct HashTable {          // device layout inside the caller-provided buffer
  HashEntry* entries;       // [cap]
  int* status;              // [1] != 0: a coordinate did not fit the packed key (v3d_hash_status)
  unsigned mask;            // cap - 1 (cap = power of two)
};

// the packed key holds 16 bits per field: batch in [0, 65535], coordinates in [-kGuard, 65535 - 2 kGuard] so that the
// +-1 voxel probes of the neighbour tables cannot wrap either
constexpr int kCoordMax = 65535 - 2 * kGuard;

__device__ __forceinline__ int hash_find(const HashTable& t, unsigned long long key) {
  unsigned slot = hash_u64(key) & t.mask;
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  for (unsigned probe = 0; probe <= t.mask; ++probe) {
    const u32x4_ e = *reinterpret_cast<const u32x4_*>(t.entries + slot);
    const unsigned long long k = ((unsigned long long)e.y << 32) | e.x;
    if (k == key) return (int)e.z;
    if (k == kEmpty) return -1;
    slot = (slot + 1) & t.mask;
  }
  return -1;
}


// load factor <= 1/4: a lookup of an ABSENT key (the common case when the fused decoder probes the corners of hypothesis points off
// the surface) runs to the first empty slot, and a wave waits for the longest of its 64 x 6 chains: at a load of 0.46 the probing
// rounds cost the decoder 27 k cycles per tile, a quarter of its time
inline unsigned table_capacity(int n) {
  unsigned cap = 64;
  while (cap < 4u * (unsigned)(n > 0 ? n : 1)) cap <<= 1;
  return cap;
}

inline HashTable table_view(void* buf, int n) {
  HashTable t;
  const unsigned cap = table_capacity(n);
  t.entries = (HashEntry*)buf;
  t.status = (int*)((char*)buf + (size_t)cap * sizeof(HashEntry));
  t.mask = cap - 1;
  return t;
}

}  // namespace v3dhash
