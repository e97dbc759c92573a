"""CPU oracle for the audio-depth hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (audio-depth-estimation_amd/) never imports it and fails
loudly when the HIP extension is missing.

Every function restates one piece of the reference (Kang-ChangWoo/audio-depth-estimation)
with plain torch-CPU / numpy ops and cites the reference file:line it follows.  The
restatement is pinned against golden vectors generated from the reference's own modules
(tests/golden/make_golden.py, run in the build container where /root/reference exists).

Pinning status:
  * unet_oracle / loss_oracle / optim_oracle / metrics_oracle: PINNED by tests/golden/*.npz
    (outputs of the reference's models/unetbaseline_model.py, utils_loss.py,
    utils_criterion.py and of torch.optim / clip_grad_norm_ as train.py calls them).
  * frontend_oracle: PARITY UNPINNED against torchaudio/torchvision (neither is installed
    and the reference pins no version); pinned only against torch.stft / F.interpolate,
    the functions those libraries delegate to, plus known-answer tests.
"""
