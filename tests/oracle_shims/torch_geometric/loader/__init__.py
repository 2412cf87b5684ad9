class DataLoader:  # never iterated by the golden generator
    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
