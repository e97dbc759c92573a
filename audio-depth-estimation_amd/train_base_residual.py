"""MI355X counterpart of /root/reference/train_base_residual.py: same flags, same loop, fused libadn steps (train_dc.py).

    python -m audio_depth_estimation_amd.train_base_residual --synthetic 64 --epochs 1 --batch_size 8
"""
from .train_dc import main_base_residual as main

if __name__ == '__main__':
    main()
