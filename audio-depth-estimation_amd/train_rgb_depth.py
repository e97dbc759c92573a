"""MI355X counterpart of /root/reference/train_rgb_depth.py: same flags, same loop, fused libadn steps (see train_dc.py).

    python -m audio_depth_estimation_amd.train_rgb_depth --synthetic 64 --nb_epochs 1 --batch_size 8
"""
from .train_dc import main_rgb as main

if __name__ == '__main__':
    main()
