"""MI355X-native radiosonde demod + FEC path: ctypes binding of libsonde_mi355.so (include/sonde_abi.h).

    _lib      the raw C ABI (argtypes / restypes), constants, structures
    batch     SondeBatch (many 48 kS/s channels), SondeChannelizer (10 MS/s -> 512 bins), SondeVfo (VFO-rate front-end)
    node      SondeNode: the one-process node-level host (libsonde_rccl.so, include/sonde_node.h): one batch per GPU, RCCL scatter of IQ rows
    shard     channel sharding for a rank-per-GPU host on plain torch.distributed (range arithmetic, scatter, frame gather)
    synth     synthetic signal generator for all seven sonde types (tests and bench; independent of the decoders' code)

Nothing here computes on the CPU: every entry point needs the HIP library and a GPU (no fallback)."""
