// Counter-based generator for on-device reset (DESIGN.md "Reset generator").
//
// The reference draws from Julia's Random (SR:120,124,128; utils.jl:24,28), whose
// seed -> stream mapping is not stable across Julia releases, so the batched engine owns
// its stream: every draw is a pure function of (seed, global agent id, episode, draw
// index).  splitmix64's finaliser (Steele, Lea, Flood 2014) is the mixing function.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RCW_HD __host__ __device__ __forceinline__
#else
#define RCW_HD inline
#endif

RCW_HD uint64_t rcw_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}

RCW_HD uint64_t rcw_episode_key(uint64_t seed, uint64_t global_agent, uint64_t episode)
{
    const uint64_t k = rcw_mix64(seed + 0x9e3779b97f4a7c15ULL * (global_agent + 1));
    return rcw_mix64(k ^ (episode * 0xd1b54a32d192ed03ULL));
}

RCW_HD uint64_t rcw_draw(uint64_t key, uint64_t n)
{
    return rcw_mix64(key + 0x9e3779b97f4a7c15ULL * (n + 1));
}

// uniform integer on 0..range-1 = high 64 bits of u * range
RCW_HD uint64_t rcw_below(uint64_t u, uint64_t range)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(u, range);
#else
    return (uint64_t)(((unsigned __int128)u * (unsigned __int128)range) >> 64);
#endif
}
