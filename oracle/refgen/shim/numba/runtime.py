class _RT:
    def get_allocation_stats(self):
        return None
rtsys = _RT()
