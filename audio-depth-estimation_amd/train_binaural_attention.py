"""MI355X counterpart of /root/reference/train_binaural_attention.py: same flags, same loop, fused libadn steps (see train_dc.py).

    python -m audio_depth_estimation_amd.train_binaural_attention --synthetic 64 --nb_epochs 1 --batch_size 8
"""
from .train_dc import main_binaural as main

if __name__ == '__main__':
    main()
