from .dataset import Dataset
from .save import save
from .general import flatten, check_symmetric
