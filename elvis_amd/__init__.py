"""elvis_amd - MI355X-native implementation of ELVIS's client-side restoration hot path.

Drop-in for the reference's `elvis.py` / `utils.py` call surface: the names
below keep the reference's spelling, argument meaning and error behaviour; the arithmetic runs
as hand-written HIP (gfx950) kernels in `elvis_amd/lib/libelvis_amd.so` behind the C ABI of
`include/elvis_amd.h`.  There is no CPU fallback: without a ROCm GPU (or without the built
library) the restoration entry points raise RuntimeError.
"""
from .sharding import (ChunkSpec, chunk_for_devices, parallel_process_frames, rank_frame_range,  # noqa: F401
                       resolve_device_list, _resolve_device_list)
from .recompose import (combine_blocks_into_image, split_image_into_blocks,  # noqa: F401
                        restore_video_adaptively)
from .tiler import (adaptive_restore, blended_restoration, resource_aware_restore,  # noqa: F401
                    _extract_tile_with_halo, extract_tile_with_halo)
from .frameio import (clear_directory, decode_strength_maps_from_npz, encode_strength_maps_to_npz,  # noqa: F401
                      get_frame_paths, load_block_masks, load_frame, load_strength_maps, save_block_masks, save_frame)
from .degrade import filter_frame_dct, filter_frame_downsample, filter_frame_gaussian  # noqa: F401
from .metrics import calculate_block_ssim, calculate_mse, calculate_psnr, masked_mse, masked_psnr  # noqa: F401
from .drivers import restore_blur_adaptive, restore_dct_adaptive, restore_downsampled_with_sinsr  # noqa: F401
from .restore import (get_sinsr_model, get_sinsr_upsample_fn, restore_frames_blur,  # noqa: F401
                      restore_frames_dct, restore_frames_rounds, restore_frames_sinsr,
                      restore_with_sinsr_naive)

__version__ = "0.1.0"
