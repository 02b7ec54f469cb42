"""MI355X-native (gfx950) inference engine for the DBNet++ -> SVTRv2 -> CTC invoice OCR hot path.

Importing the package is cheap and CPU-safe; ``DBNetPP`` / ``SVTRv2`` load ``lib/libocrvi.so`` on first use and
raise if it is missing (there is no CPU fallback)."""
from .vocab import VOCAB, Tokenizer  # noqa: F401


def __getattr__(name):
    if name == "DBNetPP":
        from .det import DBNetPP
        return DBNetPP
    if name == "SVTRv2":
        from .rec import SVTRv2
        return SVTRv2
    raise AttributeError(name)
