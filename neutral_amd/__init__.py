"""neutral_amd -- MI355X-native over-particle transport path of UoB-HPC/neutral.

Only the hot path behind the reference's ``neutral_interface.h`` lives here:

* ``csrc/``      hand-written HIP kernels (gfx950) + the C-ABI (libneutral_hip.so)
* ``host/``      plain-C host layer (deck reader, mesh, problem set-up, driver)
* ``interface``  ctypes mirror of the three interface functions
* ``host``       ctypes mirror of the host layer
* ``decks``      the four standard problem decks as data
* ``cs_table``   the cross-section table shipped with the reference, as data

Importing the package does not load the HIP library; ``neutral_amd.interface``
does, and fails loudly when it has not been built.
"""

__all__ = ["decks", "cs_table", "host", "interface", "shard"]
