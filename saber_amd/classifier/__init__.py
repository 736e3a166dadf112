"""Domain-expert classifier filter on the engine's image embeddings (reference package: saber/classifier)."""
