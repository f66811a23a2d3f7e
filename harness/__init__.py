"""Measurement harness: a small Levenberg-Marquardt pose-graph solver standing in
for ceres::Solve (Ceres is not installed here or on the GPU box).  It exists to
measure "full pose-graph solve ms" and to check final-pose parity with the SAME
solver driving the GPU backend and the CPU-oracle backend.  Not part of the
product: the reference keeps Ceres as the outer solver."""
