"""MI355X-native view-synthesis loss (Monodepth2-style) behind the reference's
``layers.py`` / ``Trainer.compute_losses`` call surface.

Sub-modules
    synth         deterministic synthetic KITTI-shaped inputs
    _abi          ctypes binding of ``libvsloss.so`` (the C-ABI boundary)
    ops           autograd Functions over the C-ABI
    layers        drop-in ``disp_to_depth / BackprojectDepth / Project3D / SSIM / get_smooth_loss``
    trainer_loss  ``generate_images_pred / compute_reprojection_loss / compute_losses`` counterparts
"""
__version__ = "0.1.0"
