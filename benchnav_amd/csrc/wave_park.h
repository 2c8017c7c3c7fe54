// wave_park.h -- the one-wave kernel's register block for parked controls (rollout_wave.inc).
//
// A lane keeps the clamped controls of its rollout's first kParkSteps steps in vector registers v[BN_PARK_BASE ..) until the
// rollout's cost -- their weight -- is known.  The block is addressed with the VGPR index mode: between s_set_gpr_idx_on (mode
// bit 3 = destination, bit 0 = source 0; index = an SGPR) and s_set_gpr_idx_off a v_mov_b32 names v[base + i] and the hardware
// adds M0[7:0].  The compiler knows nothing of values living there, so three things keep it away from the block:
//   * the kernels that use it carry amdgpu_num_vgpr(BN_PARK_BASE / 2).  On gfx90a+ (unified VGPR / AGPR file) the attribute is a
//     budget for BOTH halves: the allocator may use vector registers up to twice the number -- requested as BN_PARK_BASE itself it
//     capped nothing, and the first kernels built that way overwrote parked controls with address temporaries (found by the check
//     below before any test ran; the `reserved registers' diagnostic of the store statement is the confirmation that the block lies
//     outside what the allocator may touch, and is switched off here for that reason);
//   * the store statement names every register of the block as clobbered: the code object's register count covers it;
//   * tests/test_build_artifacts.py disassembles the shipped kernels and fails if any instruction outside these two statements
//     touches v[BN_PARK_BASE ..], or if the cap made a kernel spill.
// 128 registers in all: four waves per SIMD, i.e. the 4096 workgroups of a 256-instance launch resident at once -- at three
// waves per SIMD the same launch takes 25 % longer (profiles/r5_experiments/occupancy.txt), more than parking saves.  68 is the
// lowest base at which no variant of the kernel spills (the tail it carries as its aux workgroup needs 63-67 registers).
#pragma once
#define BN_PARK_BASE 68
#define BN_PARK_STR2(x) #x
#define BN_PARK_STR(x) BN_PARK_STR2(x)
#define BN_PARK_REG(i) "v[" BN_PARK_STR(BN_PARK_BASE) "+" #i "]"
#define BN_PARK_CLOBBERS "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"

namespace bn {

namespace {

constexpr int kParkSteps = 30;        // 60 registers: v68 .. v127

// controls of steps t, t + 1 (idx = 2 t, t even) of this lane's rollout into the block
__device__ __forceinline__ void park_store4(int idx, float a, float b, float c, float d)
{
    asm volatile("s_set_gpr_idx_on %4, 0x8\n\t"
                 "v_mov_b32 " BN_PARK_REG(0) ", %0\n\t"
                 "v_mov_b32 " BN_PARK_REG(1) ", %1\n\t"
                 "v_mov_b32 " BN_PARK_REG(2) ", %2\n\t"
                 "v_mov_b32 " BN_PARK_REG(3) ", %3\n\t"
                 "s_set_gpr_idx_off"
                 : : "v"(a), "v"(b), "v"(c), "v"(d), "s"(idx) : BN_PARK_CLOBBERS, "m0");   /* s_set_gpr_idx_on writes M0 (index and mode bits): whoever uses M0 next must set it up again */
}

// ... and back (volatile: ordered behind the stores)
__device__ __forceinline__ void park_load4(int idx, float u[4])
{
    asm volatile("s_set_gpr_idx_on %4, 0x1\n\t"
                 "v_mov_b32 %0, " BN_PARK_REG(0) "\n\t"
                 "v_mov_b32 %1, " BN_PARK_REG(1) "\n\t"
                 "v_mov_b32 %2, " BN_PARK_REG(2) "\n\t"
                 "v_mov_b32 %3, " BN_PARK_REG(3) "\n\t"
                 "s_set_gpr_idx_off"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "s"(idx) : "m0");
}

}  // namespace

}  // namespace bn

#pragma clang diagnostic pop
