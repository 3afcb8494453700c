// fi_workers.h -- persistent helper threads for fi_assemble's level chains.  A std::thread per level and assemble costs its
// creation and, worse, the HIP runtime's per-thread set-up at the thread's first call, on the critical path of every step; the
// threads here are created once and sleep between assembles.  A task runs on exactly one worker; wait() returns when it is done.
#pragma once

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace fi {

class Worker {
public:
	Worker() : thread_([this]() { loop(); }) {}
	~Worker()
	{
		{
			std::lock_guard<std::mutex> g(m_);
			stop_ = true;
		}
		cv_.notify_all();
		if (thread_.joinable()) { thread_.join(); }
	}
	void run(std::function<void()> f)  // (tasks do not throw: fi_assemble's `guarded` wrappers)
	{
		{
			std::lock_guard<std::mutex> g(m_);
			task_ = std::move(f);
			busy_ = true;
		}
		cv_.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> g(m_);
		done_cv_.wait(g, [this]() { return !busy_; });
	}

private:
	void loop()
	{
		std::unique_lock<std::mutex> g(m_);
		for (;;) {
			cv_.wait(g, [this]() { return stop_ || (busy_ && task_); });
			if (stop_) { return; }
			std::function<void()> f = std::move(task_);
			task_ = nullptr;
			g.unlock();
			f();
			g.lock();
			busy_ = false;
			done_cv_.notify_all();
		}
	}
	std::mutex              m_;
	std::condition_variable cv_, done_cv_;
	std::function<void()>   task_;
	bool                    busy_ = false, stop_ = false;
	std::thread             thread_;
};

// process-wide pool: acquire() hands out an idle worker (or starts one), release() takes it back
class WorkerPool {
public:
	Worker* acquire()
	{
		std::lock_guard<std::mutex> g(m_);
		if (!idle_.empty()) {
			Worker* w = idle_.back();
			idle_.pop_back();
			return w;
		}
		all_.emplace_back(new Worker());
		return all_.back().get();
	}
	void release(Worker* w)
	{
		std::lock_guard<std::mutex> g(m_);
		idle_.push_back(w);
	}

private:
	std::mutex                           m_;
	std::vector<std::unique_ptr<Worker>> all_;
	std::vector<Worker*>                 idle_;
};

inline WorkerPool& worker_pool()
{
	static WorkerPool pool;
	return pool;
}

}  // namespace fi
