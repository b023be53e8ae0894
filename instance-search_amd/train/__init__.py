"""Training / embedding-extraction entry points of the drop-in surface (get_embeddings, main) -- MI355X build."""
