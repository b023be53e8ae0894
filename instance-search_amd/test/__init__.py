"""Command-line test mains of the drop-in surface (same flags and printed lines as the reference) -- MI355X build."""
