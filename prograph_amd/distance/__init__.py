from .hamming import hamming
from .minkowski import minkowski
from .utils import clean_input
