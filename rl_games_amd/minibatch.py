"""PPODataset: minibatch slicing of the flattened rollout.

Interface of `rl_games.common.datasets.PPODataset` (rl_games/common/datasets.py:7-95) - same constructor
arguments, attributes (`length`, `num_games_batch`, `last_range`, `values_dict`, `special_names`, ...) and
methods - over the env-major rollout storage of this package: minibatch `i` is the contiguous row block
`[i*mb, (i+1)*mb)` of every tensor in `values_dict` (never shuffled unless `permute=True`), RNN batches
slice whole sequences, and `update_mu_sigma` writes the new policy's mu/sigma into the rows of the last
minibatch.  All slices are zero-copy views of the buffer storage, so the fused loss kernel can also write
mu/sigma in place and HIP graphs can bake the slice addresses in."""
import torch

_CONFIG_ERRORS = (('minibatch_size', 'Batch size must be divisible by minibatch size.'),
                  ('seq_length', 'Batch size must be divisible by sequence length.'))


class PPODataset:
    def __init__(self, batch_size, minibatch_size, is_discrete, is_rnn, device, seq_length, permute=False):
        self.special_names = ['rnn_states']        # per instance, like the reference (callers may append)
        sizes = {'minibatch_size': minibatch_size, 'seq_length': seq_length}
        for key, message in _CONFIG_ERRORS:
            if batch_size % sizes[key]:
                raise ValueError(message)
        self.batch_size, self.minibatch_size, self.seq_length = batch_size, minibatch_size, seq_length
        self.device = device
        self.is_rnn, self.is_discrete, self.is_continuous = is_rnn, is_discrete, not is_discrete
        self.length = batch_size // minibatch_size                 # minibatches per pass
        self.num_games_batch = minibatch_size // seq_length        # sequences per RNN minibatch
        self.permute = permute
        self.permutation_indices = (torch.arange(batch_size, dtype=torch.long, device=device) if permute else None)
        self.values_dict = None
        self.last_range = (0, 0)

    def update_values_dict(self, values_dict):
        self.values_dict = values_dict

    def _rows(self, idx):
        """[start, end) of minibatch idx; RNN minibatches are whole sequences (datasets.py:62-66)."""
        per = self.num_games_batch * self.seq_length if self.is_rnn else self.minibatch_size
        return idx * per, (idx + 1) * per

    def update_mu_sigma(self, mu, sigma):
        """datasets.py:33-43: the rows of the minibatch served last take the new policy's mu / sigma."""
        rows = slice(*self.last_range)
        for name, new in (('mu', mu), ('sigma', sigma)):
            self.values_dict[name][rows] = new

    def apply_permutation(self):
        """datasets.py:45-55: one random row order for every (non-special) entry; never for RNN data."""
        if not self.permute or self.is_rnn:
            return
        order = self.permutation_indices = torch.randperm(self.batch_size, device=self.device, dtype=torch.long)
        for name in [k for k in self.values_dict if k not in self.special_names]:
            entry = self.values_dict[name]
            if isinstance(entry, dict):
                for sub in entry:
                    entry[sub] = entry[sub][order]
            elif entry is not None:
                self.values_dict[name] = entry[order]

    def __getitem__(self, idx):
        start, end = self.last_range = self._rows(idx)
        item = {}
        for name, entry in self.values_dict.items():
            if name in self.special_names:
                continue
            if entry is None:
                if self.is_rnn:
                    item[name] = None          # RNN batches keep the key (rnn_masks: None)
            elif isinstance(entry, dict):
                item[name] = {k: t[start:end] for k, t in entry.items()}
            else:
                item[name] = entry[start:end]
        if self.is_rnn:
            g0, g1 = start // self.seq_length, end // self.seq_length
            item['rnn_states'] = [s[:, g0:g1, :].contiguous() for s in self.values_dict['rnn_states']]
        return item

    def __len__(self):
        return self.length

    def __iter__(self):
        return (self[i] for i in range(self.length))
