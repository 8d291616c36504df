"""torch Dataset over {tensor -> label} (prograph/utils/dataset.py of the reference)."""
import torch


class Dataset(torch.utils.data.Dataset):
    def __init__(self, list_IDs, labels):
        self.list_IDs = list_IDs
        self.labels = labels

    def __len__(self):
        return len(self.list_IDs)

    def __add__(self, other):
        return Dataset(list(self.list_IDs) + list(other.list_IDs), {**self.labels, **other.labels})

    def __getitem__(self, index):
        key = self.list_IDs[index]
        return key, self.labels[key]
