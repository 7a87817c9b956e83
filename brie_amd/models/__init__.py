from .engine import BRIE2
from .wrap import BRIE_RV, concate, fit_BRIE_matrix, fitBRIE
from .simulator import simulator
