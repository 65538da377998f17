"""Restatement of the handful of diffusers==0.29.2 symbols the reference UNet imports
(requirements.txt:2; import sites audio_cond_unet_3d_condition.py:25-30, utils.py:8,
ff_spatio_audio_temp_transformer_3d.py:10-16).  diffusers is NOT installed in this image and there is
no network, so these are written from the library's published behaviour on top of torch built-ins
(F.scaled_dot_product_attention, F.gelu, nn.Linear ...).  TEST INFRASTRUCTURE ONLY: used by
oracle/gen_golden.py in the build container to import /root/reference/avgen/models/unets and dump
golden vectors.  Nothing under asva_amd/ imports it and it never runs in the product path.
"""
__version__ = "0.29.2-restated"
