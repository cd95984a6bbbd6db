"""MI355X-native PDP (propagation / decimation / prediction) SAT-solving hot path.

Keeps the module layout of the reference's ``pdp`` package (reference: src/pdp/) for the inference
path so that user code written against it keeps working; the compute lives in hand-written HIP
kernels behind the C ABI of include/pdp_hip.h (see pdp/native.py).
"""
