from torch.utils.data import Dataset, DataLoader, IterableDataset  # noqa: F401
