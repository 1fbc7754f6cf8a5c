"""Binding stubs a maintainer of the reference adds to switch its hot path to the MI355X library:
`adet_C` = the `adet._C` op module (third_party/adet/layers/csrc/vision.cpp:52-55), `d2_register` = the META_ARCH class
Detectron2's `build_model(cfg)` constructs.  Shipped as code and tested (tests/test_compat_cpu.py, tests/test_compat_gpu.py)."""
