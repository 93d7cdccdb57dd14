// Chip geometry: the ONE place it lives.  Everything in csrc/ that depends on how
// many compute units the device has, how much LDS a CU holds or how much of it one workgroup can
// be given reads a DeviceGeometry — queried from the HIP runtime once per device
// (hipDeviceGetAttribute), not compiled in.  On a partitioned MI355X (CPX / DPX: 32 / 128 CUs per
// logical GPU) the LDS budgets, the "a SIMD for every wavefront" limits of the helper-wavefront
// kernels and the measured batch-size thresholds follow the logical device.
//
// The batch-size thresholds of the kernel choice (select_fused, i2lqr_recommended_layout, the
// scheduling options of the lane kernels) were MEASURED on the full chip (256 CUs).  They are
// occupancy effects — where a launch stops leaving every wavefront a SIMD of its own, where it
// starts to fill the chip — so on a device with another CU count they are scaled by cus / 256
// (scaled() / scaled_from()): an estimate, exact at 256.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace i2lqr {

struct DeviceGeometry {
  int cus = 256;                        // MI355X: 8 XCDs x 32 CUs
  int simds_per_cu = 4;                 // CDNA
  size_t lds_per_cu = 160 * 1024;       // gfx950
  size_t max_dyn_lds = 160 * 1024;      // most dynamic LDS one workgroup can be given (opt-in)
  size_t default_dyn_lds = 64 * 1024;   // ... without hipFuncSetAttribute
  int wave = 64;
  int faked = 0;                        // I2LQR_FAKE_CUS is in effect (debug override of `cus`)
  int queried = 0;                      // 1: from the runtime; 0: no device visible, MI355X figures

  static constexpr int kRefCUs = 256;   // the chip the batch thresholds were measured on

  int64_t simds() const { return (int64_t)cus * simds_per_cu; }
  // problems of a one-problem-per-lane launch with one wavefront on every SIMD
  int64_t full_batch() const { return simds() * wave; }
  // a batch size measured on the 256-CU chip, on this device
  int64_t scaled(int64_t measured) const {
    if (cus == kRefCUs) return measured;
    const int64_t v = (measured * cus + kRefCUs - 1) / kRefCUs;
    return v < 1 ? 1 : v;
  }
  // ... for "first batch size at which the other side wins" entries (last loser + 1)
  int64_t scaled_from(int64_t from) const { return cus == kRefCUs ? from : scaled(from - 1) + 1; }
  // LDS a wavefront can count on when one wavefront sits on every SIMD of its CU, less the
  // allocation granularity (gfx950: 160 / 4 - 4 = 36 KiB)
  size_t lds_per_simd_wave() const { return lds_per_cu / (size_t)simds_per_cu - 4 * 1024; }
};

// Geometry of the CURRENT device (cached per device); the MI355X figures above when no device is
// visible (host-only calls such as i2lqr_recommended_layout in a build container).
const DeviceGeometry& device_geometry();

}  // namespace i2lqr
