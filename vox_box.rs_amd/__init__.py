"""vox_box.rs_amd -- MI355X-native (gfx950) batched replacement for the per-frame DSP hot
path of the Rust crate vox_box 0.3.0.

The product is the C-ABI shared library ``lib/libvoxbox_hip.so`` (include/voxbox_hip.h) built
from the hand-written HIP kernels in ``csrc/``.  This Python module is only a ctypes binding
of that ABI (used by tests/ and bench.py); the C++ mirror of the crate's trait surface is
``host/voxbox.hpp``.  There is NO CPU fallback: if the library is missing or no gfx950 device
is visible, construction fails loudly.

The directory name contains a dot, so import it through ``__graft_entry__.load_package()``
(importlib, module name ``vox_box_rs_amd``).
"""
from .voxbox import (  # noqa: F401
    VoxBox, VoxBoxError, DeviceArray, LIB_PATH, load_library, exported_symbols,
    window_table, frame_count, hz_to_mel, mel_to_hz, poly_degree, poly_off_low,
    WINDOW_HANNING, WINDOW_HANNING_LAG, WINDOW_HANNING_PERIODIC, WINDOW_RECTANGLE,
    MALE_FORMANT_ESTIMATES, FEMALE_FORMANT_ESTIMATES,
    FRAME_OK, FRAME_ERR_LPC, FRAME_ERR_POLYNOMIAL, FRAME_ERR_NAN, FRAME_ERR_PANIC,
    AnalysisParams, Comm, comm_unique_id, comm_live_count, gather_plan, shard_range, shard_samples,
    ShardPlan, shard_plan, shard_local_segments, mfcc_bins,
    GATHER_NONE, GATHER_RECV, GATHER_SEND, GATHER_COPY,
    MAX_PITCH_CANDIDATES, pitch_max_candidates,
)
from . import shard  # noqa: F401
