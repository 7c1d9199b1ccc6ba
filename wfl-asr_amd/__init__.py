"""MI355X-native drop-in for the WFL-ASR inference hot path (16 kHz audio -> BIO tags -> .lab).

Product code lives here; `oracle/` is test infrastructure and is never imported from this package.
"""
__version__ = "0.1.0"
