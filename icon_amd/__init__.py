"""icon_amd - MI355X-native implicit-surface query engine for ICON (gfx950, HIP).

Public surface (mirrors the reference's query interface, see icon_amd/engine.py):
    IconQueryEngine.query      == HGPIFuNet.query        (lib/net/HGPIFuNet.py:268-367)
    query_func                 == lib/common/train_util.py:324-348
    DenseReconEngine           == the reconEngine object (lib/common/seg3d_lossless.py:36)
The compute path is libicon_amd.so (include/icon_amd.h); there is no CPU fallback.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # lazy: `import icon_amd` must not need torch or the shared library
    if name in ("IconQueryEngine", "MeshHandle", "FeatHandle", "MlpHandle", "query_func"):
        from . import engine
        return getattr(engine, name)
    if name in ("DenseReconEngine", "slab_bounds", "export_mesh_numpy"):
        from . import recon
        return getattr(recon, name)
    raise AttributeError(name)
