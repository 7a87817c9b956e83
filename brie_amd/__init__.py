"""brie_amd -- MI355X-native brie-quant inference core (one hot path, no CPU fallback).

Public surface = the reference's `brie.models` names
(/root/reference/brie/models/__init__.py:1-2): BRIE2, fit_BRIE_matrix, fitBRIE
(+ the result object BRIE_RV).
"""
from .version import __version__
from .models import BRIE2, BRIE_RV, fit_BRIE_matrix, fitBRIE, concate

# names the reference exposes at package level (brie/__init__.py): brie.read_npz, brie.read_h5ad, brie.match,
# brie.pp.filter_genes, brie.io.dump_results
from . import io
from . import preprocessing as pp
from .io import read_npz, read_h5ad
from .preprocessing import match
