"""fastqpacker_amd — MI355X-native per-block FASTQ codec (fqpack hot path).

Thin ctypes binding over libfqzhip.so (include/fqz.h).  There is no CPU
fallback: importing works anywhere, but every call needs the HIP library and
a GPU and raises FqzError otherwise.
"""
from ._lib import (FqzError, Ctx, lib, build, library_path,  # noqa: F401
                   ENCODING_PHRED33, ENCODING_PHRED64, DETECT_ENCODING, BATCH_FINAL, BATCH_V3, BATCH_SEG, BATCH_HALVES, DEFAULT_BLOCK_SIZE,
                   STREAM_NAMES, Options, DecompressOptions)
from . import encoder, compress, fqformat  # noqa: F401
# fastqpacker_amd.sharding needs torch.distributed and is imported on demand

