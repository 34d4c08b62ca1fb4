"""uni-adapter_amd: MI355X-native (gfx950) implementation of Uni-Adapter's per-sample test-time
hot path - the Uni3D point-cloud encoder forward and the MODE-DOTA cache step - behind the
reference's own Python call surface (DOTA_mix, Uni3D.encode_pc, test_zeroshot_3d_core).

Every device operation goes through the C-ABI library `libuniadapter_hip.so` (hand-written HIP,
see include/uniadapter.h); there is no CPU or PyTorch fallback: importing `uni_adapter_amd.lib`
without the built library raises.
"""
__version__ = '0.1.0'
