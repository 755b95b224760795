"""PPODataset: minibatch slicing of the flattened rollout.

Mirror of `rl_games.common.datasets.PPODataset` (rl_games/common/datasets.py:7-95): minibatch
`i` is the contiguous row block `[i*mb, (i+1)*mb)` of every tensor in `values_dict` (env-major
rows, never shuffled unless `permute=True`), RNN batches slice whole sequences, and
`update_mu_sigma` writes the new policy's mu/sigma into the rows of the last minibatch.  All
slices are zero-copy views, so the fused loss kernel can also write mu/sigma in place."""
import torch


class PPODataset:
    def __init__(self, batch_size, minibatch_size, is_discrete, is_rnn, device, seq_length, permute=False):
        if batch_size % minibatch_size != 0:
            raise ValueError('Batch size must be divisible by minibatch size.')
        if batch_size % seq_length != 0:
            raise ValueError('Batch size must be divisible by sequence length.')
        self.is_rnn = is_rnn
        self.seq_length = seq_length
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size
        self.device = device
        self.length = batch_size // minibatch_size
        self.is_discrete = is_discrete
        self.is_continuous = not is_discrete
        self.num_games_batch = minibatch_size // seq_length
        self.special_names = ['rnn_states']
        self.permute = permute
        self.values_dict = None
        self.last_range = (0, 0)
        if permute:
            self.permutation_indices = torch.arange(batch_size, dtype=torch.long, device=device)

    def update_values_dict(self, values_dict):
        self.values_dict = values_dict

    def update_mu_sigma(self, mu, sigma):
        start, end = self.last_range
        self.values_dict['mu'][start:end] = mu
        self.values_dict['sigma'][start:end] = sigma

    def apply_permutation(self):
        if self.permute and not self.is_rnn:
            perm = torch.randperm(self.batch_size, device=self.device, dtype=torch.long)
            self.permutation_indices = perm
            for key, value in self.values_dict.items():
                if key in self.special_names or value is None:
                    continue
                if isinstance(value, dict):
                    for k, v in value.items():
                        value[k] = v[perm]
                else:
                    self.values_dict[key] = value[perm]

    def _span(self, idx):
        """Row range [start, end) of minibatch idx and, for RNN data, its sequence range."""
        if self.is_rnn:
            g0 = idx * self.num_games_batch
            g1 = g0 + self.num_games_batch
            return g0 * self.seq_length, g1 * self.seq_length, (g0, g1)
        start = idx * self.minibatch_size
        return start, start + self.minibatch_size, None

    def __getitem__(self, idx):
        start, end, games = self._span(idx)
        self.last_range = (start, end)

        def cut(v):
            return {k: t[start:end] for k, t in v.items()} if isinstance(v, dict) else v[start:end]
        item = {}
        for name, v in self.values_dict.items():
            if name in self.special_names:
                continue
            if v is None:
                if self.is_rnn:
                    item[name] = None          # RNN batches keep the key (rnn_masks: None)
                continue
            item[name] = cut(v)
        if games is not None:
            item['rnn_states'] = [s[:, games[0]:games[1], :].contiguous() for s in self.values_dict['rnn_states']]
        return item

    def __len__(self):
        return self.length

    def __iter__(self):
        return (self[i] for i in range(self.length))
