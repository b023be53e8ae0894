"""Network wrappers and custom modules of the drop-in surface, backed by libisx on the GPU -- MI355X build."""
