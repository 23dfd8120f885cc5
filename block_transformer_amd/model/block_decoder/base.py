"""Block decoder base: constructor contract of the reference's BaseBlockDecoder (model/block_decoder/base.py:9-28).
The block-decoding losses (:26-74) are training-only and out of the decode path."""


class BaseBlockDecoder:
    def __init__(self, hidden_size: int, n_embedding_tokens=8, use_block_decoding_loss=False,
                 block_decoding_loss_type="contrastive"):
        self.hidden_size = hidden_size
        self.n_embedding_tokens = n_embedding_tokens
        self.use_block_decoding_loss = use_block_decoding_loss
        self.block_decoding_loss_type = block_decoding_loss_type
