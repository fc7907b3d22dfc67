def find_dotenv(*a, **k): return ''
def load_dotenv(*a, **k): return False
