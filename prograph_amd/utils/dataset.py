"""
Key/label dataset for `Prograph.gen_dataloaders` (the reference's prograph/utils/dataset.py):
item i is `(keys[i], labels[keys[i]])`; two datasets concatenate with `+`.
"""
from torch.utils import data as _data


class Dataset(_data.Dataset):
    def __init__(self, list_IDs, labels):
        self.list_IDs, self.labels = list_IDs, labels

    def __getitem__(self, index):
        key = self.list_IDs[index]
        return key, self.labels[key]

    def __len__(self):
        return len(self.list_IDs)

    def __add__(self, other):
        merged = dict(self.labels)
        merged.update(other.labels)
        return Dataset([*self.list_IDs, *other.list_IDs], merged)
