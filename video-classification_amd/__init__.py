"""MI355X-native SlowFast training path (drop-in for zc402/video-classification's train.py + model/my_slowfast.py).

Layout:
  csrc/        hand-written gfx950 HIP kernels + the C ABI of include/sfk.h  -> libsfk.so
  build.py     compiles libsfk.so with hipcc (cross-compiles without a GPU)
  _lib.py      ctypes binding of include/sfk.h (fails loudly when the library is missing)
  plan.py      conv geometry -> implicit-GEMM pass descriptors (forward, data-gradient classes, filter gradient)
  arch.py      SlowFast wiring (reference geometry and canonical 8x8) as a flat layer list
  engine.py    buffers, forward/backward schedule, parameter arena, checkpoints
  slowfast.py  init_my_slowfast / slowfast_r50_8x8 model facades (reference model/my_slowfast.py surface)
  config.py    yacs-compatible CfgNode + the reference's config/defaults.py keys
  train.py     ModelManager / Trainer (reference train.py surface)
  dist.py      one-process-per-GPU gradient all-reduce over RCCL
"""
__version__ = "0.1.0"
