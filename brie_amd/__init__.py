"""brie_amd -- MI355X-native brie-quant inference core (one hot path, no CPU fallback).

Public surface = the reference's `brie.models` names
(/root/reference/brie/models/__init__.py:1-2): BRIE2, fit_BRIE_matrix, fitBRIE
(+ the result object BRIE_RV).
"""
from .version import __version__
from .models import BRIE2, BRIE_RV, fit_BRIE_matrix, fitBRIE, concate
