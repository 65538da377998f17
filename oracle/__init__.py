"""CPU oracle for the AVSyncD denoising path — TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the timed CPU baseline.  Nothing under asva_amd/ imports it; the product path
raises when libavsd_hip.so is missing rather than falling back to anything here.

What it is: a plain PyTorch fp32 (CPU) restatement of the reference algorithm, each function citing
the reference file:line it follows:
  unet_ref.py      AudioUNet3DConditionModel.forward and everything below it
                   (avgen/models/unets/**; U1-U13 of SURVEY.md §8a)
  vae_ref.py       AutoencoderKL.decode as called at pipeline_audio_cond_animation.py:206-213 (V1)
  sched_ref.py     PNDMScheduler / DDIMScheduler as called at :325-327,337,364 (S1)
  audio_ref.py     waveform -> log-mel (avgen/data/utils.py:26-55) and ImageBindSegmaskAudioEncoder.forward
                   (segmask_imagebind.py:80-123); ImageBind / torchaudio absent -> PARITY UNPINNED
  pipeline_ref.py  latent preparation, denoising loop, guidance, first-frame pinning, decode
                   post-processing (:234-261, :330-375; P1, P3)

Pinning status:
  * The reference ships NO tests, golden vectors or fixtures.
  * unet_ref.py is pinned against the reference's OWN UNet code imported in the build container
    (oracle/gen_golden.py; tests/golden/*.pt hold the resulting input/output vectors):
    restatement == reference to ~1e-6 rel-L2 in fp32.
  * The arithmetic of the third-party primitives the reference calls — diffusers==0.29.2
    Attention/AttnProcessor2_0, FeedForward/GEGLU, Timesteps/TimestepEmbedding, AutoencoderKL,
    PNDMScheduler, DDIMScheduler (requirements.txt:2; not vendored, not installed, no network) — is
    restated from the published definitions: PARITY UNPINNED by any reference-side vector for those;
    they are pinned instead against torch built-ins and closed-form known answers
    (tests/test_oracle.py).
"""
