// dbg.hpp -- the ONLY compile-time switches of the library.  The product build defines none of them; dfu3d_amd/_build.py
// lists the test / timing builds that do (VARIANTS).  Nothing here changes results except where it says so.
//
//   DFU3D_DBG_COMBO_KEYBITS=<n>  test build: the packed min-(key | pixel) word of the voxel table keeps only n key bits, so keys
//                                collide all the time and the exact repair of k_bp_vox / k_ovf_* / k_bp_fix (practically never
//                                taken in the product) does the work.  Same results as the product.
//   DFU3D_DBG_NO_MID             test build: no middle tier of the bin classification -- everything float32 leaves undecided takes
//                                the full fp64 kernel (k_bp_bin_amb).  Same results as the product.
//   DFU3D_DBG_TIMING             dev build: cycles of thread 0 between the DBG_T(k) marks of a kernel, summed over workgroups;
//                                every source file that uses the marks has its own counters and exports a reader with
//                                DBG_T_READER(name) (pixel_stage.hip: dfu3d_debug_timing_pixel; tools/p1_timing.py).
//                                Same results as the product.
#pragma once

#ifdef DFU3D_DBG_COMBO_KEYBITS
constexpr int DBG_COMBO_KEYBITS = DFU3D_DBG_COMBO_KEYBITS;
#else
constexpr int DBG_COMBO_KEYBITS = 64;
#endif

#ifdef DFU3D_DBG_NO_MID
constexpr bool DBG_NO_MID = true;
#else
constexpr bool DBG_NO_MID = false;
#endif

#ifdef DFU3D_DBG_TIMING
constexpr int DBG_T_SLOTS = 32;
static __device__ unsigned long long g_dbg_cycles[DBG_T_SLOTS];
#define DBG_T_START() long long dbg_t_ = clock64(); const long long dbg_t0_ = dbg_t_; (void)dbg_t0_
#define DBG_T(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); atomicAdd(&g_dbg_cycles[k], (unsigned long long)(t_ - dbg_t_)); dbg_t_ = t_; } } while (0)
#define DBG_T_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_dbg_cycles[k], 1ull); } while (0)
#define DBG_T_ADD(k, v) do { if (threadIdx.x == 0) atomicAdd(&g_dbg_cycles[k], (unsigned long long)(v)); } while (0)
#define DBG_T_MAX(k, v) do { if (threadIdx.x == 0) atomicMax(&g_dbg_cycles[k], (unsigned long long)(v)); } while (0)
// at file scope, outside any namespace: int name(unsigned long long out[DBG_T_SLOTS], int reset)
#define DBG_T_READER(name)                                                                                              \
  extern "C" int name(unsigned long long *out, int reset) {                                                            \
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_cycles), sizeof(unsigned long long) * DBG_T_SLOTS) != hipSuccess)     \
      return DFU3D_ELAUNCH;                                                                                             \
    if (reset) {                                                                                                        \
      unsigned long long z[DBG_T_SLOTS] = {0};                                                                          \
      if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_cycles), z, sizeof(z)) != hipSuccess) return DFU3D_ELAUNCH;                \
    }                                                                                                                   \
    return DFU3D_OK;                                                                                                    \
  }
#else
#define DBG_T_START() do {} while (0)
#define DBG_T(k) do {} while (0)
#define DBG_T_COUNT(k) do {} while (0)
#define DBG_T_ADD(k, v) do {} while (0)
#define DBG_T_MAX(k, v) do {} while (0)
#define DBG_T_READER(name)
#endif
