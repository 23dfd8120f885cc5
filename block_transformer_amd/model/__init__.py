from .block_decoder.gpt_neo_x import GPTNeoXBlockDecoder
from .block_transformer import BlockTransformer
from .embedder.lookup import LookupConfig, LookupEmbedder
from .neox import GenerationConfig, GPTNeoXConfig
from .token_decoder.gpt_neo_x import GPTNeoXTokenDecoder
from .utils import (load_block_decoder_from_config, load_block_transformer_from_config, load_embedder_from_config,
                    load_token_decoder_from_config)
