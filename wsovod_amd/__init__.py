"""wsovod_amd -- MI355X-native (gfx950) implementation of WSOVOD's per-image detection hot path.

Layout mirrors the reference package for the path only:
  csrc/      hand-written HIP kernels + the C-ABI (include/wsovod_hip.h)
  layers/    autograd fronts of the native ops       (reference: wsovod/layers)
  modeling/  backbone, poolers, roi_heads, class_heads, meta_arch (reference: wsovod/modeling)
  engine/    run_step / DDP / optimizer              (reference: wsovod/engine)
"""
__version__ = "0.1.0"
