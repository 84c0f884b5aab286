"""Timing-only build of the LDS-shared greedy kernel for the occupancy experiment (profiles/r05_occupancy_timing_experiment.txt):
patches iiv_greedy.hip in place so that -DIIV_EXP_ALIAS aliases the MT19937 look-ahead onto the block and the two offsets' L1
halves in LDS (wrong tie-breaks and table values, the same instruction stream) -- then e.g.
    python tools/exp_alias_patch.py ii-vision_amd/csrc/iiv_greedy.hip && tools/build_variant.sh w12alias "-DIIV_EXP_ALIAS -DIIV_SHARED_W=12"
and `git checkout ii-vision_amd/csrc/iiv_greedy.hip` afterwards.  Never part of the product build."""
import sys
p=sys.argv[1]
s=open(p).read()
def rep(old,new):
    global s
    assert old in s, old
    s=s.replace(old,new,1)
rep("""    uint32_t mt[624 + 256];     // random's current MT19937 block + the first 256 words of the next one
};""","""#ifdef IIV_EXP_ALIAS
    uint32_t mt[624];
#else
    uint32_t mt[624 + 256];     // random's current MT19937 block + the first 256 words of the next one
#endif
};""")
rep("    static constexpr int kL1Bytes = kOffsets * kHalfBytes;","""#ifdef IIV_EXP_ALIAS
    static constexpr int kL1Bytes = kHalfBytes;
#else
    static constexpr int kL1Bytes = kOffsets * kHalfBytes;
#endif""")
rep("for (int i = threadIdx.x; i < kQuads; i += 64 * W) dst[h * kQuads + i] = src[i];","""for (int i = threadIdx.x; i < kQuads; i += 64 * W)
#ifdef IIV_EXP_ALIAS
                dst[i] = src[i];
#else
                dst[h * kQuads + i] = src[i];
#endif""")
rep("        const uint32_t sl_d = (kOddInLds ? (uint32_t)SC::kHalfBytes : l1_d) + (split_content_left<MODE>(c, 1) << (T::kLeftRowBits + 1));","""#ifdef IIV_EXP_ALIAS
        const uint32_t sl_d = (kOddInLds ? 0u : l1_d) + (split_content_left<MODE>(c, 1) << (T::kLeftRowBits + 1));
#else
        const uint32_t sl_d = (kOddInLds ? (uint32_t)SC::kHalfBytes : l1_d) + (split_content_left<MODE>(c, 1) << (T::kLeftRowBits + 1));
#endif""")
rep("    uint32_t *ahead = mt + 624;","""#ifdef IIV_EXP_ALIAS
    uint32_t *ahead = mt;
#else
    uint32_t *ahead = mt + 624;
#endif""")
open(p,'w').write(s)
