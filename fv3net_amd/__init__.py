"""MI355X (gfx950) implementation of fv3net's column-wise ML tendency inference and
cubed-sphere coarse-graining hot path.

The compute lives in ``libfv3hip.so`` (hand-written HIP, C ABI in ``include/fv3hip.h``); this
package is the thin Python host side that mirrors the reference's own interfaces:

========================  ==========================================================
``fv3net_amd.cubedsphere``  ``vcm.cubedsphere`` coarsening functions and ``regridz``
``fv3net_amd.mappm``        the f2py module ``mappm`` (``mappm.mappm(...)``)
``fv3net_amd.thermo``       the three ``vcm.calc.thermo`` pressure helpers on the path
``fv3net_amd.fit``          ``fv3fit`` Predictor API, io registry, dense predictor
``fv3net_amd.emulation``    ``emulation`` microphysics hook
``fv3net_amd.ops``          array-level entry points (torch device tensors in / out)
========================  ==========================================================

PyTorch is used for device memory, streams and ``torch.distributed`` only.  There is no CPU
fallback: importing the compute entry points without the built extension raises.
"""

__version__ = "0.1.0"
