"""One rank of the two-rank dataset-driver test (tests/test_multirank_gpu.py): launched by torch.distributed.run inside the
scratch tree the parent wrote; calls the reference's `generate_videos_for_dataset` (12 keyword arguments,
pipeline_audio_cond_animation.py:472-485) and records what THIS rank wrote."""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import asva_amd.pipeline as P
    import avgen.pipelines.pipeline_audio_cond_animation as ref_api
    from asva_amd import dist as adist

    steps, frames, fps, size, nclips = (int(v) for v in sys.argv[1:6])
    rank, local_rank, world = adist.env_rank_world()
    P.AudioCondAnimationPipeline.generation_steps = steps          # the driver hard-codes 50 (:442)
    written = []
    real_writer = P.write_video

    def tap(filename, video_array, fps_, audio_array=None, audio_fps=16000, audio_codec="aac"):
        written.append({"file": filename, "sha": hashlib.sha256(video_array.contiguous().cpu().numpy().tobytes()).hexdigest(),
                        "shape": list(video_array.shape)})
        return real_writer(filename, video_array, fps_, audio_array, audio_fps, audio_codec)

    P.write_video = tap
    # the unchanged reference script passes torch.device("cuda") on every rank
    ref_api.generate_videos_for_dataset(exp_root="exp", checkpoint=7, dataset="AVSync15", image_size=(size, size), video_fps=fps,
                                        video_num_frame=frames, num_clips_per_video=nclips, audio_guidance_scale=4.0,
                                        text_guidance_scale=1.0, random_seed=0, device=torch.device("cuda"), dtype=torch.float32)
    with open(f"rank{rank}.json", "w") as f:
        json.dump({"rank": rank, "world": world, "device": torch.cuda.current_device(), "written": written}, f)


if __name__ == "__main__":
    main()
