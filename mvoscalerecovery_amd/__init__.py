"""mvoscalerecovery_amd — MI355X (gfx950) implementation of the per-frame scale-recovery hot path
of TimingSpace/MVOScaleRecovery behind the reference's ``ScaleEstimator`` call surface.

    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

Layout: ``csrc/`` HIP kernels + C ABI (include/mvosr.h) -> ``libmvosr.so``; ``_lib`` ctypes
binding; ``packing`` HBM layout; ``engine`` launch wrappers; ``scale_calculator`` the drop-in
class; ``offline`` the main_offline-shaped driver loop; ``sharding`` multi-GPU frame sharding;
``synth`` synthetic KITTI-shaped inputs.
"""
__version__ = "0.1.0"
