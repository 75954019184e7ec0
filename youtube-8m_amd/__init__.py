"""yt8m_amd -- MI355X-native hot path of wangheda/youtube-8m behind the reference's plugin surface.

The directory is called ``youtube-8m_amd`` (not an importable identifier); load it through
``__graft_entry__.load_package()`` which registers it as the package ``yt8m_amd``.

Module names mirror /root/reference/youtube-8m-wangheda/: ``models``, ``video_level_models``,
``frame_level_models``, ``losses``, ``model_utils``, ``utils``, ``feature_transform``, ``eval_util``,
``average_precision_calculator``, ``mean_average_precision_calculator``, ``train``.
Compute goes through ``libyt8m_hip.so`` (C ABI in include/yt8m_hip.h); there is NO CPU fallback.
"""
__version__ = "0.1.0"
